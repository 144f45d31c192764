#!/usr/bin/env python3
"""Per-shape timing of the bf16-operand GEMM (subgc_gemm_bf16) on the contractions of one Full_GC_Kar / Flickr train step,
against torch.mm on bf16 tensors (hipBLASLt: a yardstick only, the product never calls it).

    python tools/gemm_bf16_bench.py [--config full_gc_kar|flickr|kar] [--only substr]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sub-gc_amd"))
import torch  # noqa: E402

from subgc import ops  # noqa: E402

BF = torch.bfloat16


def shapes(cfg):
    if cfg == "full_gc_kar":
        S, T, R, E, A, V1, NN, NK, D, L = 1280, 17, 1000, 1000, 512, 9488, 9472, 16640, 2048, 1024
    elif cfg == "flickr":
        S, T, R, E, A, V1, NN, NK, D, L = 320, 17, 1000, 1000, 512, 7001 + 7, 6464, 19264, 4096, 2048
    else:
        S, T, R, E, A, V1, NN, NK, D, L = 640, 17, 1000, 1000, 512, 9488, 4736, 8320, 2048, 1024
    TS = S * T
    return [
        ("fwd att_lstm step", "nt", S, 4 * R, 2 * R, T), ("fwd lang_lstm step", "nt", S, 4 * R, 3 * R, T), ("fwd h2att step", "nt", S, A, R, T),
        ("bwd dH2 step", "nn", S, 3 * R, 4 * R, T), ("bwd dH1 step", "nn", S, 2 * R, 4 * R, T), ("bwd dh1<-att step", "nn", S, R, A, T),
        ("fwd Gx all t", "nt", TS, 4 * R, E, 1), ("fwd logits all t", "nt", TS, V1, R, 1),
        ("bwd dHout", "nn", TS, R, V1, 1), ("bwd dxt", "nn", TS, E, 4 * R, 1),
        ("dW logit", "tn", V1, R, TS, 1), ("dW lang ih", "tn", 4 * R, 2 * R, TS, 1), ("dW lstm 4Rx1R", "tn", 4 * R, R, TS, 4),
        ("dW h2att", "tn", A, R, TS, 1), ("fusion obj_v_proj", "nt", NN, L, D, 1), ("gcn fc_lft", "nt", NK, 512, L, 4),
        ("gcn fc_rgt", "nt", NK, L, 512, 4), ("bwd gcn fc_lft dx", "nn", NK, L, 512, 4), ("dW gcn fc_rgt", "tn", L, 512, NK, 4),
        ("att_embed", "nt", S * 36, R, L, 1), ("dW att_embed", "tn", R, L, S * 36, 1),
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
    ap.add_argument("--config", default="full_gc_kar")
    ap.add_argument("--only", default="")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--shape", default="")
    ap.add_argument("--pad", type=int, default=0, help="row pitch of every operand rounded up to this many elements (64 = 128-byte aligned rows)")
    ap.add_argument("--out16", action="store_true", help="bf16 destination (what torch.mm on bf16 tensors writes) instead of fp32")
    ap.add_argument("--tile", default="0", choices=("0", "128", "256", "p8"), help="force the workgroup form (SUBGC_GEMM_TILE128 / TILE256 / TILE_P8 flag bits)")
    ap.add_argument("--no-p8", action="store_true", help="forbid the eight-phase form (SUBGC_GEMM_NO_P8)")
    ap.add_argument("--splits", type=int, default=0, help="force this many K parts (SUBGC_GEMM_SPLITS(n); 0 = the cost model decides)")
    a = ap.parse_args()
    ops.WS_MBYTES = max(ops.WS_MBYTES, 1024)
    ops.gemm_tune.b16_bits = ops.gemm_tune.TILE_BITS[a.tile if a.tile == "p8" else int(a.tile)] | ((1 << 14) if a.no_p8 else 0) | ((a.splits & 15) << 8)
    sh = shapes(a.config)
    if a.shape:
        sh = [("custom",) + tuple(int(x) if i else x for i, x in enumerate(s.split(","))) + (1,) for s in a.shape.split(";")]
    dev = "cuda:0"
    tot_mine = tot_lib = tot_flop = 0.0
    print(f"{'shape':22s} {'mode':4s} {'M':>6s} {'N':>6s} {'K':>6s} {'x':>3s} {'subgc us':>9s} {'TF/s':>7s} {'torch us':>9s} {'TF/s':>7s}")
    for name, mode, M, N, K, cnt in sh:
        if a.only and a.only not in name:
            continue
        g = torch.Generator().manual_seed(0)
        x = torch.randn(M, K, generator=g).to(dev).to(BF)
        w = torch.randn(K, N, generator=g).to(dev).to(BF)

        def padded(t):                                    # same values, rows `--pad`-aligned (a view with ld > columns)
            if not a.pad:
                return t
            ld = (t.size(1) + a.pad - 1) // a.pad * a.pad
            buf = torch.zeros(t.size(0), ld, device=dev, dtype=t.dtype)
            buf[:, :t.size(1)] = t
            return buf[:, :t.size(1)]
        x, w = padded(x), padded(w)
        out16 = torch.empty(M, N, device=dev, dtype=BF)
        out = torch.empty(M, N, device=dev, dtype=BF) if a.out16 else torch.empty(M, N, device=dev)
        if mode == "nt":
            wt = padded(w.t().contiguous())
            mine = lambda: ops.gemm(x, wt, out, tb=True)
            lib = lambda: torch.mm(x, wt.t(), out=out16)
        elif mode == "nn":
            mine = lambda: ops.gemm(x, w, out)
            lib = lambda: torch.mm(x, w, out=out16)
        else:
            xt = padded(x.t().contiguous())
            mine = lambda: ops.gemm(xt, w, out, ta=True)
            lib = lambda: torch.mm(xt.t(), w, out=out16)
        tm, tl = timeit(mine, a.iters), timeit(lib, a.iters)
        fl = 2.0 * M * N * K
        tot_mine += tm * cnt; tot_lib += tl * cnt; tot_flop += fl * cnt
        print(f"{name:22s} {mode:4s} {M:6d} {N:6d} {K:6d} {cnt:3d} {tm*1e6:9.1f} {fl/tm/1e12:7.1f} {tl*1e6:9.1f} {fl/tl/1e12:7.1f}")
    print(f"per-step total: subgc {tot_mine*1e3:.2f} ms ({tot_flop/tot_mine/1e12:.1f} TF/s)   torch.mm {tot_lib*1e3:.2f} ms ({tot_flop/tot_lib/1e12:.1f} TF/s)")


if __name__ == "__main__":
    main()
