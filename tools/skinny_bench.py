#!/usr/bin/env python3
"""Weight-streaming GEMM of the one-image decode step (M = kept sub-graphs <= 16): us per launch and effective TB/s for the
four per-step matrices, cycled in step order so that the 120 MB of weights see the cache state a real decode step leaves."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "sub-gc_amd"), ROOT]
import torch
from subgc import ops

dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 10
shapes = [("att-LSTM [h|h]", 4000, 2000), ("h2att", 512, 1000), ("lang-LSTM", 4000, 3000), ("logit", 9488, 1000)]
g = torch.Generator(device="cpu").manual_seed(0)
ws = [torch.randn(n, k, generator=g).to(dev) for _, n, k in shapes]
xs = [torch.randn(M, k, generator=g).to(dev) for _, n, k in shapes]
ys = [torch.empty(M, n, device=dev) for _, n, k in shapes]
for w, x, y in zip(ws, xs, ys):
    ops.gemm(x, w, y, tb=True)
    ref = x.double() @ w.double().t()
    err = float((y.double() - ref).abs().max() / ref.abs().max())
    assert err < 1e-5, err
reps = 200
ev = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in shapes] for _ in range(reps)]
for r in range(reps):
    for i, (w, x, y) in enumerate(zip(ws, xs, ys)):
        ev[r][i][0].record(); ops.gemm(x, w, y, tb=True); ev[r][i][1].record()
torch.cuda.synchronize()
tot = 0.0
for i, (name, n, k) in enumerate(shapes):
    t = sorted(ev[r][i][0].elapsed_time(ev[r][i][1]) for r in range(20, reps))
    us = 1e3 * t[len(t) // 2]
    tot += us
    print(f"{name:16s} N={n:5d} K={k:5d}  {us:7.2f} us  {4e-6 * n * k / us:6.2f} TB/s")
print(f"per step {tot:.1f} us (M={M})")
