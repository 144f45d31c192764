#!/usr/bin/env python3
"""The eight-phase bf16 GEMM form against the ring forms and torch.mm (hipBLASLt, yardstick only) on the large products of the bf16 train
steps, plus a K sweep on the logit-shaped product that separates the K-loop pace from the per-tile cost.

    python tools/p8_probe.py [--iters 20] [--sweep]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sub-gc_amd"))
import torch  # noqa: E402

from subgc import ops  # noqa: E402

ops.WS_MBYTES = 2048

BF = torch.bfloat16
DEV = "cuda:0"
SHAPES = [("nt", 14593, 9488, 1000, 32), ("nt", 14593, 4000, 1000, 32), ("nn", 14593, 1000, 9488, 32), ("nn", 14593, 1000, 4000, 32),
          ("tn", 9488, 1000, 14593, 32), ("tn", 4000, 2000, 14593, 32), ("tn", 4000, 1000, 14593, 32),
          ("nt", 16640, 1024, 1024, 32), ("nn", 16640, 1024, 1024, 32), ("tn", 1024, 1024, 16640, 32), ("nt", 16640, 512, 1024, 16), ("nt", 16640, 1024, 512, 16),
          ("nt", 9472, 1024, 2048, 32), ("nt", 9472, 1024, 1024, 16), ("nt", 1280, 4000, 3000, 32), ("nt", 1280, 4000, 2000, 32), ("nn", 1280, 3000, 4000, 32),
          ("nt", 8192, 8192, 8192, 32), ("nt", 4096, 4096, 4096, 32),
          ("nt", 3706, 7008, 1000, 32), ("nt", 6464, 2048, 4096, 32), ("nt", 19264, 2048, 512, 16), ("nn", 19264, 2048, 1024, 32), ("tn", 2048, 4096, 6464, 32)]


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
    return a.elapsed_time(b) / iters * 1e3


def operands(mode, M, N, K, pad):
    g = torch.Generator().manual_seed(M + N + K)

    def mk(r, c):
        ld = (c + pad - 1) // pad * pad if pad else c
        buf = torch.zeros(r, ld, dtype=BF, device=DEV)
        buf[:, :c] = torch.randn(r, c, generator=g).to(DEV).to(BF)
        return buf[:, :c]
    a = mk(K, M) if mode == "tn" else mk(M, K)
    b = mk(N, K) if mode == "nt" else mk(K, N)
    return a, b


COLD = {"on": False, "buf": None}


def timeit_cold(fn, iters):
    """each timed call behind a 1 GiB fill: operands come from HBM, not from the Infinity Cache the previous call left them in"""
    if COLD["buf"] is None:
        COLD["buf"] = torch.empty(1 << 30, dtype=torch.uint8, device=DEV)
    tot = 0.0
    for i in range(iters + 2):
        COLD["buf"].fill_(i & 255)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        if i >= 2:
            tot += a.elapsed_time(b)
    return tot / iters * 1e3


def run(mode, M, N, K, pad, iters, out16, forms):
    global timeit
    if COLD["on"]:
        timeit = timeit_cold
    a, b = operands(mode, M, N, K, pad)
    out = torch.empty(M, N, device=DEV, dtype=BF if out16 else torch.float32)
    o16 = torch.empty(M, N, device=DEV, dtype=BF)
    res = {}
    ref = None
    for name, kw in forms:
        with ops.gemm_tune(**kw):
            try:
                ops.gemm(a, b, out, ta=mode == "tn", tb=mode == "nt")
            except ops.SubgcError:
                res[name] = float("nan")                     # a forced split the workspace cannot hold
                continue
            if ref is None:
                ref = out.clone()
            else:
                err = float((out.float() - ref.float()).abs().max()) / max(float(ref.float().abs().max()), 1e-9)
                if err > (1e-2 if out16 else 1e-4):
                    name = name + "!ERR%.1e" % err
            res[name] = timeit(lambda: ops.gemm(a, b, out, ta=mode == "tn", tb=mode == "nt"), iters)
    am = a.t() if mode == "tn" else a
    bm = b.t() if mode == "nt" else b
    res["torch"] = timeit(lambda: torch.mm(am, bm, out=o16), iters)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--sweep", action="store_true")
    ap.add_argument("--out16", action="store_true")
    ap.add_argument("--only", default="")
    ap.add_argument("--cold", action="store_true", help="flush the caches (1 GiB fill) before every timed call")
    a = ap.parse_args()
    COLD["on"] = a.cold
    forms = [("auto", dict(no_p8=True)), ("r256", dict(tile=256)), ("p8", dict(tile="p8")), ("p8x2", dict(tile="p8", splits=2)), ("p8x4", dict(tile="p8", splits=4))]
    print("# us per launch (TF/s); fp32 destination unless --out16; `pad` = row pitch multiple in elements")
    print(f"{'mode':4s} {'M':>6s} {'N':>6s} {'K':>6s} {'pad':>3s} " + " ".join(f"{n:>14s}" for n, _ in forms) + f" {'torch(bf16 out)':>16s}")
    for mode, M, N, K, pad in SHAPES:
        if a.only and a.only not in f"{mode},{M},{N},{K}":
            continue
        fl = 2.0 * M * N * K
        r = run(mode, M, N, K, pad, a.iters, a.out16, forms)
        cells = []
        for n, _ in forms:
            key = [k for k in r if k.split("!")[0] == n][0]
            cells.append(f"{r[key]:7.1f} ({fl / r[key] / 1e6:5.0f}){'!' if '!' in key else ' '}")
        print(f"{mode:4s} {M:6d} {N:6d} {K:6d} {pad:3d} " + " ".join(cells) + f" {r['torch']:8.1f} ({fl / r['torch'] / 1e6:5.0f})")
    if a.sweep:
        print("# K sweep, nt 14592 x 9472 (57 x 37 tiles of 256 x 256 = 8.24 rounds), bf16 destination: us per launch")
        forms2 = [("r256", dict(tile=256)), ("p8", dict(tile="p8"))]
        for K in (64, 128, 256, 512, 1024, 2048, 4096):
            r = run("nt", 14592, 9472, K, 64, max(5, a.iters // 2), True, forms2)
            print(f"K={K:5d} " + " ".join(f"{k}:{v:8.1f}" for k, v in r.items()))
        print("# K sweep, nt 8192 x 8192 (exactly 4 rounds), bf16 destination")
        for K in (256, 1024, 4096, 8192):
            r = run("nt", 8192, 8192, K, 64, max(5, a.iters // 2), True, forms2)
            print(f"K={K:5d} " + " ".join(f"{k}:{v:8.1f}" for k, v in r.items()))


if __name__ == "__main__":
    main()
