#!/usr/bin/env python3
"""Do a layer's weight-gradient product and the backward's data-gradient chain overlap when the former runs on a side stream?  n x (dx = dy W
on the main stream, dW = dy^T x [+ bias sum] on 0 / 1 / 2 side streams), GCN shapes of Full_GC_Kar.  Stand-alone: bf16 1072 -> 765 us for 12
pairs, fp32 4113 -> 3704; inside the real step the gain does not materialise (tools/overlap_report.py)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sub-gc_amd"))
from subgc import ops  # noqa: E402

DEV = "cuda:0"
K, M, N, n = 16640, 1024, 512, 12
for store in ("bf16", "f32"):
    dt = torch.bfloat16 if store == "bf16" else torch.float32
    dys = [torch.randn(K, M, device=DEV).to(dt) for _ in range(n)]
    xs = [torch.randn(K, N, device=DEV).to(dt) for _ in range(n)]
    dWs = [torch.zeros(M, N, device=DEV) for _ in range(n)]
    dbs = [torch.zeros(M, device=DEV) for _ in range(n)]
    Ws = [torch.randn(M, N, device=DEV).to(dt) for _ in range(n)]
    dxs = [torch.zeros(K, N, device=DEV, dtype=dt) for _ in range(n)]
    main = torch.cuda.current_stream()
    sides = [torch.cuda.Stream(), torch.cuda.Stream()]
    for s in sides:
        with torch.cuda.stream(s):
            ops.ensure_workspace(torch.device(DEV))

    def serial(chain):
        for i in range(n):
            if chain:
                ops.gemm(dys[i], Ws[i], dxs[i])
            ops.wgrad(dys[i], xs[i], dWs[i], dbs[i], accum=True, db_accum=True)

    def forked(chain, ns):
        for i in range(n):
            s = sides[i % ns]
            s.wait_stream(main)
            with torch.cuda.stream(s):
                ops.wgrad(dys[i], xs[i], dWs[i], dbs[i], accum=True, db_accum=True)
            if chain:
                ops.gemm(dys[i], Ws[i], dxs[i])
        for s in sides[:ns]:
            main.wait_stream(s)

    def timed(fn, *a):
        for _ in range(2):
            fn(*a)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn(*a)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 5 * 1e3

    for chain in (False, True):
        print(store, "with the dx chain" if chain else "weight gradients only", f"serial {timed(serial, chain):8.1f} us   1 side stream {timed(forked, chain, 1):8.1f}"
              f"   2 side streams {timed(forked, chain, 2):8.1f}")
