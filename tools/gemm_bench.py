#!/usr/bin/env python3
"""Per-shape fp32 GEMM timing on the MI355X: subgc_gemm_f32 vs torch.mm (hipBLASLt/rocBLAS, yardstick
only -- the product never calls it).  Shapes = the distinct contractions of one Sub_GC_Kar train step.

    python tools/gemm_bench.py [--only recurrent]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sub-gc_amd"))
import torch  # noqa: E402

from subgc import ops  # noqa: E402

S, T, R, E, A, V1 = 640, 17, 1000, 1000, 512, 9488
TS = S * T
# (name, mode, M, N, K, launches per step)
SHAPES = [
    ("fwd att_lstm step", "nt", S, 4 * R, 2 * R, T), ("fwd lang_lstm step", "nt", S, 4 * R, 3 * R, T), ("fwd h2att step", "nt", S, A, R, T),
    ("bwd dH2 step", "nn", S, 3 * R, 4 * R, T), ("bwd dH1 step", "nn", S, 2 * R, 4 * R, T), ("bwd dh1<-att step", "nn", S, R, A, T),
    ("fwd Gx all t", "nt", TS, 4 * R, E, 1), ("fwd logits all t", "nt", TS, V1, R, 1),
    ("bwd dHout", "nn", TS, R, V1, 1), ("bwd dxt", "nn", TS, E, 4 * R, 1),
    ("dW logit", "tn", V1, R, TS, 1), ("dW lang ih", "tn", 4 * R, 2 * R, TS, 1), ("dW lstm 4Rx1R", "tn", 4 * R, R, TS, 4),
    ("dW h2att", "tn", A, R, TS, 1), ("fusion obj_v_proj", "nt", 4736, 1024, 2048, 1), ("gcn fc_lft", "nt", 8320, 512, 1024, 2),
    ("gcn fc_rgt", "nt", 8320, 1024, 512, 2), ("gpn_fc0", "nt", 2560, 512, 2048, 1), ("fc_embed0", "nt", S, 2048, 2048, 1),
]


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--shape", default="", help="extra shape mode,M,N,K (e.g. nt,4096,4096,8192); replaces the built-in list")
    ap.add_argument("--mode", default="f32", choices=("f32", "bf16x3", "bf16"), help="arithmetic of the 128x128-tile forms (ops.gemm_mode)")
    a = ap.parse_args()
    ops.set_gemm_mode(a.mode)
    global SHAPES
    if a.shape:
        SHAPES = [("custom",) + tuple(int(x) if i else x for i, x in enumerate(sh.split(","))) + (1,) for sh in a.shape.split(";")]
    dev = "cuda:0"
    ops.ensure_workspace(dev)
    tot_mine = tot_lib = tot_flop = 0.0
    print(f"{'shape':22s} {'mode':4s} {'M':>6s} {'N':>6s} {'K':>6s} {'x':>3s} {'subgc us':>9s} {'TF/s':>7s} {'torch us':>9s} {'TF/s':>7s}")
    for name, mode, M, N, K, cnt in SHAPES:
        if a.only and a.only not in name:
            continue
        g = torch.Generator().manual_seed(0)
        x = torch.randn(M, K, generator=g).to(dev)
        w = torch.randn(K, N, generator=g).to(dev)
        out = torch.empty(M, N, device=dev)
        if mode == "nt":
            wt = w.t().contiguous()
            mine = lambda: ops.gemm(x, wt, out, tb=True)
            lib = lambda: torch.mm(x, wt.t(), out=out)
        elif mode == "nn":
            mine = lambda: ops.gemm(x, w, out)
            lib = lambda: torch.mm(x, w, out=out)
        else:
            xt = x.t().contiguous()
            mine = lambda: ops.gemm(xt, w, out, ta=True)
            lib = lambda: torch.mm(xt.t(), w, out=out)
        tm, tl = timeit(mine, a.iters), timeit(lib, a.iters)
        fl = 2.0 * M * N * K
        tot_mine += tm * cnt; tot_lib += tl * cnt; tot_flop += fl * cnt
        print(f"{name:22s} {mode:4s} {M:6d} {N:6d} {K:6d} {cnt:3d} {tm*1e6:9.1f} {fl/tm/1e12:7.1f} {tl*1e6:9.1f} {fl/tl/1e12:7.1f}")
    print(f"per-step total: subgc {tot_mine*1e3:.2f} ms ({tot_flop/tot_mine/1e12:.1f} TF/s)   torch.mm {tot_lib*1e3:.2f} ms ({tot_flop/tot_lib/1e12:.1f} TF/s)")


if __name__ == "__main__":
    main()
