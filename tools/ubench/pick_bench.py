#!/usr/bin/env python3
"""subgc_decode_pick alone: time per launch against the row count (greedy on raw logits, top-k 3), rows x 9488 fp32 logits."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "sub-gc_amd"), ROOT]
import torch
from subgc import ops
dev = torch.device("cuda:0")
V, T = 9488, 20
for n in (10, 100, 900, 2560, 8192):
    logits = torch.randn(n, V, device=dev)
    seq = torch.zeros(n, T, dtype=torch.long, device=dev); slp = torch.zeros(n, T, device=dev)
    it = torch.zeros(n, dtype=torch.long, device=dev); unf = torch.zeros(n, dtype=torch.int32, device=dev)
    cnt = torch.zeros(T, dtype=torch.int32, device=dev); u = torch.rand(n, device=dev)
    for k in (0, 3):
        def run():
            ops.decode_pick(logits, k, 0.6 if k else 1.0, u if k else None, 0, seq, slp, it, unf, cnt[0:1], None, raw=True)
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(50):
            run()
        b.record(); torch.cuda.synchronize()
        us = 1e3 * a.elapsed_time(b) / 50
        print(f"rows {n:5d} k={k}: {us:7.1f} us  ({n * V * 4 / us / 1e6:7.2f} TB/s)")
