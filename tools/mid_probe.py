"""Phase times of the fused middle of a decoder step (csrc/recurrent_mid.hip) on the three train configs' step shapes:
in-kernel wall-clock stamps (100 MHz) per workgroup at the phase boundaries + the launch's event time, against the three launches it replaces.

    python tools/mid_probe.py [--bwd]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "sub-gc_amd"))
from subgc import ops  # noqa: E402

SHAPES = {  # name: (rows, R, A, mean set length, max set length, bf16, split-K parts of the gate product)
    "kar_f32_m640": (640, 1000, 512, 6.6, 11, False, 3),
    "kar_f32_m320": (320, 1000, 512, 6.6, 11, False, 3),
    "full_gc_bf16_m1280": (1280, 1000, 512, 37, 37, True, 3),
    "full_gc_bf16_m640": (640, 1000, 512, 37, 37, True, 3),
    "flickr_bf16_m320": (320, 1000, 512, 20, 101, True, 3),
}


def time_it(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--flags", type=int, default=0)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    for name, (S, R, A, lmean, lmax, bf, parts) in SHAPES.items():
        gen = np.random.default_rng(1)
        rnd = lambda *s, sc=1.0: torch.from_numpy((gen.standard_normal(s) * sc).astype(np.float32)).to(dev)
        lens_h = np.clip(gen.poisson(lmean, size=S), 1, lmax).astype(np.int32) if lmean != lmax else np.full(S, lmax, np.int32)
        lens = torch.from_numpy(lens_h).to(dev)
        off = torch.zeros(S, dtype=torch.int32)
        off[1:] = torch.cumsum(torch.from_numpy(lens_h)[:-1], 0)
        off = off.to(dev)
        total = int(lens_h.sum())
        planes = rnd(parts, S, 4 * R, sc=0.5)
        g1, g2, b0, b1 = rnd(S, 4 * R, sc=0.3), rnd(S, 4 * R, sc=0.3), rnd(4 * R, sc=0.1), rnd(4 * R, sc=0.1)
        c_prev = rnd(S, R)
        wq, bq, w_a, b_a = rnd(A, R, sc=R ** -0.5), rnd(A, sc=0.1), rnd(A, sc=0.3), rnd(1)
        u, v = rnd(total, A), rnd(total, R)
        wq_mid = ops.transpose_f32(wq)
        if bf:
            wq, u, v = ops.as_b16(wq), ops.as_b16(u), ops.as_b16(v)
            wq_mid = wq
        H = ops.act_padded((S, 3 * R), dev, bf, zero_rows=S)
        Hn = ops.act_padded((S, 2 * R), dev, bf, zero_rows=S)
        c, G, q, al = torch.zeros(S, R, device=dev), torch.zeros(S, 4 * R, device=dev), torch.zeros(S, A, device=dev), torch.zeros(S, lmax, device=dev)
        QP = torch.empty(8 * S * A, device=dev)
        pre = torch.empty(S, 4 * R, device=dev)

        def three():
            ops.lstm_fwd_planes_probe(planes, parts, g1, g2, b0, b1, c_prev, c, H[:, R:2 * R], Hn[:, R:], G, S, R) if hasattr(ops, "lstm_fwd_planes_probe") else \
                ops.lstm_fwd(planes[0], g1, g2, b0, b1, c_prev, c, H[:, R:2 * R], Hn[:, R:], None, 1.0, None, G, S, R)
            nq, sq = ops.gemm_planes(H[:, R:2 * R], wq, QP, tb=True)
            ops.attn_fwd(u, v, q, w_a, b_a, off, lens, H[:, :R], al, S, A, R, q=(QP, nq, sq, bq))

        wgs = (S + max((S + 255) // 256, 1) - 1) // max((S + 255) // 256, 1)
        stamps = torch.zeros(wgs, 8, dtype=torch.int64, device=dev)

        def one(st=None):
            ops.mid_fwd(planes[0], parts, S * 4 * R, g1, g2, b0, b1, c_prev, c, H[:, R:2 * R], Hn[:, R:], G, wq_mid, bq, q, u, v, w_a, b_a, off, lens,
                        H[:, :R], al, S, R, A, flags=a.flags, stamps=st)

        t3, t1 = time_it(three), time_it(one)
        one(stamps)
        torch.cuda.synchronize()
        st = stamps.cpu().numpy().astype(np.float64) * 0.01           # 100 MHz -> us
        ph = np.diff(st[:, :6], axis=1)
        names = ["prologue+cell", "query", "q_out+scores", "softmax", "context"]
        span = (st[:, 5].max() - st[:, 0].min())
        print(f"{name}: three launches {t3:.1f} us (1-plane cell), fused {t1:.1f} us; {wgs} workgroups, in-kernel span {span:.1f} us; "
              + ", ".join(f"{n} {ph[:, i].mean():.1f} (max {ph[:, i].max():.1f})" for i, n in enumerate(names))
              + f"; workgroup start skew {st[:, 0].max() - st[:, 0].min():.1f} us")


if __name__ == "__main__":
    main()
