"""ops.wgrad (weight + bias gradient in one launch) against ops.gemm(ta=True) + ops.colsum on the train configs' shapes.
usage: python tools/wgrad_bench.py [f32|bf16]"""
import sys
import torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sub-gc_amd"))
from subgc import ops

DEV = "cuda:0"
store = sys.argv[1] if len(sys.argv) > 1 else "bf16"
SHAPES = [(16640, 1024, 512), (16640, 512, 1024), (9472, 1024, 1024), (14000, 4000, 2000), (14000, 4000, 1000), (14000, 9488, 1000),
          (14000, 512, 1000), (4736, 512, 1024), (256, 1000, 2048), (19264, 2048, 1024)]


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for K, M, N in SHAPES:
    dy, x = torch.randn(K, M, device=DEV), torch.randn(K, N, device=DEV)
    if store == "bf16":
        dy, x = dy.bfloat16(), x.bfloat16()
    dW, db = torch.zeros(M, N, device=DEV), torch.zeros(M, device=DEV)
    t_fold = timed(lambda: ops.wgrad(dy, x, dW, db, accum=True, db_accum=True))
    t_gemm = timed(lambda: ops.gemm(dy, x, dW, ta=True, accum=True))
    t_col = timed(lambda: ops.colsum(dy, out=db, accumulate=True))
    print(f"{store} rows {K:6d} out {M:5d} in {N:5d}: fold {t_fold:8.1f} us   gemm {t_gemm:8.1f} + colsum {t_col:6.1f} = {t_gemm + t_col:8.1f} us")
