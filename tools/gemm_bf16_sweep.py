#!/usr/bin/env python3
"""Does the bf16 GEMM's cost model (gemm_bf16.hip plan_for) pick the fastest form?  Records every subgc_gemm_bf16 / _wgrad call of ONE real
train step of a bf16 config (shape, leading dimensions, epilogue), then replays each distinct shape stand-alone under the plan's own choice
and under every forced (tile, K parts) form the call's epilogue allows (SUBGC_GEMM_TILE128 / TILE256 / TILE_P8 / SUBGC_GEMM_SPLITS(n) flag bits).

    python tools/gemm_bf16_sweep.py [--config full_gc_kar|flickr] [--min-us 15]
"""
import argparse
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "sub-gc_amd"), ROOT):
    sys.path.insert(0, p)
import torch  # noqa: E402

import bench  # noqa: E402
from subgc import ops, synthetic  # noqa: E402
import subgc.models as models  # noqa: E402

BF = torch.bfloat16
DEV = "cuda:0"


def timeit(fn, iters=12):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="full_gc_kar", choices=["full_gc_kar", "flickr"])
    ap.add_argument("--min-us", type=float, default=0.0)
    ap.add_argument("--top", type=int, default=3, help="how many of the fastest forced forms to list per shape")
    a = ap.parse_args()
    cfg = bench.CONFIGS[a.config]
    torch.manual_seed(1234)
    model = models.setup(argparse.Namespace(**cfg["opt"])).to(DEV).train()
    b = {k: v.to(DEV) for k, v in synthetic.make_train_batch(cfg["batch"], seed=1000, **cfg["data"]).items()}
    lw = models.LossWrapper(model, None)

    def step():
        model.flatten_grads()
        models.total_loss(lw(*bench.lw_args(b))).backward()

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    seen = collections.OrderedDict()
    real = ops.call

    def spy(name, *args):
        if name == "subgc_gemm_bf16":
            ta, tb, M, N, K, _, lda, _, ldb, c32, _, c16, _, bias, add, _, keep, _, flags, m_dev = args[:20]
            key = ("tn" if ta else "nt" if tb else "nn", M, N, K, lda, ldb, c32 is not None, c16 is not None, bias is not None, add is not None,
                   keep is not None, flags & 3, m_dev is not None, False)
            seen[key] = seen.get(key, 0) + 1
        elif name == "subgc_gemm_bf16_wgrad":
            M, N, K, _, lddy, _, ldx, _, _, _, flags, _, m_dev = args[:13]
            key = ("tn", M, N, K, lddy, ldx, True, False, False, False, False, flags & 2, m_dev is not None, True)
            seen[key] = seen.get(key, 0) + 1
        return real(name, *args)

    ops.call = spy
    step()
    torch.cuda.synchronize()
    ops.call = real
    rows = []
    for key, cnt in seen.items():
        mode, M, N, K, lda, ldb, o32, o16, hb, ha, hk, fl, ragged, wg = key
        if ragged:
            continue
        g = torch.Generator().manual_seed(0)
        A = torch.randn(*((K, lda) if mode == "tn" else (M, lda)), generator=g).to(DEV).to(BF)[:, : (M if mode == "tn" else K)]
        B = torch.randn(*((N, ldb) if mode == "nt" else (K, ldb)), generator=g).to(DEV).to(BF)[:, : (K if mode == "nt" else N)]
        out = torch.zeros(M, N, device=DEV) if o32 else torch.zeros(M, N, device=DEV, dtype=BF)
        out16 = torch.zeros(M, N, device=DEV, dtype=BF) if (o32 and o16) else None
        bias = torch.zeros(N, device=DEV) if hb else None
        add = torch.zeros(M, N, device=DEV) if ha else None
        keep = torch.ones(M, N, device=DEV, dtype=torch.uint8) if hk else None
        db = torch.zeros(M, device=DEV)

        def run():
            if wg:
                ops.wgrad(A, B, out, db, accum=bool(fl & 2), db_accum=True)
            else:
                ops.gemm(A, B, out, ta=mode == "tn", tb=mode == "nt", bias=bias, add=add, keep=keep, relu=bool(fl & 1), accum=bool(fl & 2), out16=out16)

        ops.gemm_tune.b16_bits = 0
        auto = timeit(run)
        best, best_name = auto, "auto"
        plain = not ha and not hk
        res = []
        for tile, tbit in ((128, 64), (256, 128), ("p8", 1 << 13)):
            for sp in (1, 2, 3, 4, 6, 8):
                if sp > 1 and not plain:
                    continue
                if sp > 1 and (K + 31) // 32 // sp < 12:
                    continue
                ops.gemm_tune.b16_bits = tbit | (sp << 8)
                try:
                    t = timeit(run)
                except ops.SubgcError:
                    continue
                res.append((t, f"{tile}x{sp}"))
                if t < best:
                    best, best_name = t, f"{tile}x{sp}"
        ops.gemm_tune.b16_bits = 0
        auto = min(auto, timeit(run))                      # again, warm: the first timing of a shape also pays its allocations' first touch
        if best > auto:
            best, best_name = auto, "auto"
        rows.append((cnt * auto, cnt, key, auto, best, best_name, sorted(res)[:a.top]))
    rows.sort(key=lambda r: -r[0])
    tot_auto = sum(r[1] * r[3] for r in rows)
    tot_best = sum(r[1] * r[4] for r in rows)
    print(f"{a.config}: {len(rows)} distinct non-ragged shapes, {sum(r[1] for r in rows)} calls per step; stand-alone sum {tot_auto / 1e3:.2f} ms with the plan's choices, "
          f"{tot_best / 1e3:.2f} ms with the best forced form of every shape")
    print(f"{'mode':4s} {'M':>6s} {'N':>6s} {'K':>6s} {'epi':6s} {'x':>3s} {'auto us':>8s} {'best us':>8s} {'form':>6s} {'gain/step us':>12s}   next best")
    for _, cnt, key, auto, best, name, top in rows:
        if auto < a.min_us:
            continue
        mode, M, N, K, lda, ldb, o32, o16, hb, ha, hk, fl, ragged, wg = key
        epi = ("W" if wg else "") + ("b" if hb else "") + ("+" if ha else "") + ("r" if fl & 1 else "") + ("d" if hk else "") + ("A" if fl & 2 else "") + ("h" if o16 else "")
        print(f"{mode:4s} {M:6d} {N:6d} {K:6d} {epi:6s} {cnt:3d} {auto:8.1f} {best:8.1f} {name:>6s} {cnt * (auto - best):12.1f}   " + "  ".join(f"{n}:{t:.1f}" for t, n in top))


if __name__ == "__main__":
    main()
