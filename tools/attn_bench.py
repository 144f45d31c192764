"""Stand-alone timing of the per-step attention kernels (attn_fwd / attn_bwd with deferred d(u), d(v)) on the set shapes of the
train configs.  `--copies N` rotates over N copies of the sets so that a call does not find them in the Infinity Cache.

    python tools/attn_bench.py [--config fgk|flickr|kar|all] [--copies 3] [--reps 40]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "sub-gc_amd"))
from subgc import ops  # noqa: E402

CONFIGS = {           # rows, (min, max) set length, A, R, bf16 sets
    "fgk": (1280, (37, 37), 512, 1000, True),
    "flickr": (640, (2, 30), 512, 1000, True),
    "kar": (640, (2, 11), 512, 1000, False),        # 128 images x 5 sentences
}


def run(name, copies, reps):
    S, (lo, hi), A, R, b16 = CONFIGS[name]
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(0)
    lens_h = rng.integers(lo, hi + 1, size=S).astype(np.int32)
    off_h = np.concatenate([[0], np.cumsum(lens_h)[:-1]]).astype(np.int32)
    tot = int(lens_h.sum())
    lens, off = torch.from_numpy(lens_h).to(dev), torch.from_numpy(off_h).to(dev)
    dt = torch.bfloat16 if b16 else torch.float32
    us = [(torch.randn(tot, A, device=dev) * 0.5).to(dt) for _ in range(copies)]
    vs = [torch.randn(tot, R, device=dev).to(dt) for _ in range(copies)]
    ah = torch.randn(S, A, device=dev) * 0.5
    w_a, b_a = torch.randn(A, device=dev) * 0.1, torch.zeros(1, device=dev)
    n = int(lens_h.max())
    ctx, alpha = torch.empty(S, R, device=dev, dtype=dt), torch.empty(S, n, device=dev)
    dctx = torch.randn(S, R, device=dev)
    dah, dw_a, db_a = torch.empty(S, A, device=dev), torch.empty(S, A, device=dev), torch.empty(S, device=dev)
    de_keep, dctx_keep = torch.empty(S, n, device=dev), torch.empty(S, R, device=dev)

    def timed(fn):
        # the calls are captured into one hipGraph: a Python call costs 10-12 us of host time, more than the small kernels run
        for i in range(3):
            fn(i % copies)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with ops.graph_capture(g, dev):
            for i in range(reps):
                fn(i % copies)
        g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(5):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        return 1e3 * e0.elapsed_time(e1) / reps / 5

    fwd = timed(lambda i: ops.attn_fwd(us[i], vs[i], ah, w_a, b_a, off, lens, ctx, alpha, S, A, R))
    bwd = timed(lambda i: ops.attn_bwd(us[i], vs[i], ah, w_a, off, lens, alpha, dctx, dah, None, None, dw_a, db_a, S, A, R,
                                       dctx_keep=dctx_keep, de_keep=de_keep))
    eb = 2 if b16 else 4
    mb = tot * (A + R) * eb / 1e6
    print(f"{name:7s} rows {S} nodes {tot} sets {mb:7.1f} MB copies {copies}: fwd {fwd:6.1f} us ({mb / fwd:5.2f} TB/s)  "
          f"bwd {bwd:6.1f} us ({mb / bwd:5.2f} TB/s)")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="all")
    ap.add_argument("--copies", type=int, default=3)
    ap.add_argument("--reps", type=int, default=40)
    a = ap.parse_args()
    for c in (CONFIGS if a.config == "all" else [a.config]):
        for k in sorted({1, a.copies}):
            run(c, k, a.reps)
