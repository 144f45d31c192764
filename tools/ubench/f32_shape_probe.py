import sys, torch
sys.path.insert(0, "sub-gc_amd")
from subgc import ops
dev = torch.device("cuda:0"); ops.ensure_workspace(dev)
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n
for (M, N, K, tb) in [(877, 9488, 1000, True), (877, 4000, 3000, True), (877, 4000, 2000, True), (800, 9488, 1000, True), (700, 9488, 1000, True), (4736, 512, 1024, True), (4736, 512, 1024, False), (8320, 512, 1024, True), (8320, 1024, 512, True), (640, 512, 1000, True), (640, 1000, 512, False), (2560, 512, 2048, True), (4204, 1000, 1024, True)]:
    a = torch.randn(M, K, device=dev); b = torch.randn(N, K, device=dev) if tb else torch.randn(K, N, device=dev)
    o = torch.empty(M, N, device=dev)
    d = t(lambda: ops.gemm(a, b, o, tb=tb))
    with ops.gemm_tune(no_splitk=True):
        ns = t(lambda: ops.gemm(a, b, o, tb=tb))
    print(f"{'nt' if tb else 'nn'} {M}x{N}x{K}: default {d:.1f} us ({2*M*N*K/d/1e6:.1f} TF/s), no split-K {ns:.1f} us", flush=True)
