#!/usr/bin/env python3
"""In-situ A/B of the bf16 GEMM dispatch: every subgc_gemm_bf16 / _pair / _wgrad call of a real train step is bracketed by events (at the
ops.call level, so weight gradients and pair launches are included; the products the C recurrence issues are not), once with the
library's own plan and once per extra flag set (default: SUBGC_GEMM_NO_P8), several steps each, interleaved.  Operands are as cold as
the step leaves them -- which is what the stand-alone sweep (tools/gemm_bf16_sweep.py) cannot show.

    python tools/gemm_insitu_ab.py [--config full_gc_kar|flickr] [--reps 4] [--min-us 15]
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
import subgc._lib as L  # noqa: E402
import subgc.models as models  # noqa: E402

DEV = "cuda:0"
VARIANTS = {"plan": 0, "no_p8": 1 << 14, "p8": 1 << 13, "r256": 128, "r128": 64}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="full_gc_kar", choices=["full_gc_kar", "flickr", "kar"])
    ap.add_argument("--reps", type=int, default=4)
    ap.add_argument("--min-us", type=float, default=15.0)
    ap.add_argument("--variants", default="plan,no_p8")
    a = ap.parse_args()
    names = a.variants.split(",")
    cfg = bench.CONFIGS[a.config]
    f32 = cfg["dtype"] == "f32"
    torch.manual_seed(1234)
    model = models.setup(argparse.Namespace(**cfg["opt"])).to(DEV).train()
    b = {k: v.to(DEV) for k, v in synthetic.make_train_batch(cfg["batch"], seed=1000, **cfg["data"]).items()}
    lw = models.LossWrapper(model, None)

    def step():
        model.flatten_grads()
        models.total_loss(lw(*bench.lw_args(b))).backward()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    real = ops.call
    rec = []

    def spy(name, *args):
        if name not in ("subgc_gemm_bf16", "subgc_gemm_bf16_pair", "subgc_gemm_bf16_wgrad", "subgc_gemm_f32", "subgc_gemm_f32_pair", "subgc_gemm_f32_wgrad"):
            return real(name, *args)
        if name == "subgc_gemm_bf16":
            ta, tb, M, N, K = args[:5]
            c32, c16, bias, add, keep, flags, m_dev = args[9], args[11], args[13], args[14], args[16], args[18], args[19]
            key = ("tn" if ta else "nt" if tb else "nn", M, N, K, ("f" if c32 is not None else "") + ("h" if c16 is not None else "") + ("b" if bias is not None else "") +
                   ("+" if add is not None else "") + ("d" if keep is not None else "") + ("r" if flags & 1 else "") + ("A" if flags & 2 else "") + ("~" if m_dev is not None else ""))
        elif name == "subgc_gemm_f32":
            ta, tb, M, N, K = args[:5]
            bias, add, keep, flags, a_rows, c_rows, m_dev = args[11], args[12], args[14], args[16], args[17], args[18], args[19]
            key = ("tn" if ta else "nt" if tb else "nn", M, N, K, ("b" if bias is not None else "") + ("+" if add is not None else "") + ("d" if keep is not None else "") +
                   ("r" if flags & 1 else "") + ("A" if flags & 2 else "") + ("g" if a_rows is not None else "") + ("s" if c_rows is not None else "") + ("~" if m_dev is not None else ""))
        elif name.endswith("_pair"):
            ta, tb, M, N, K = args[:5]
            key = ("tn" if ta else "nt" if tb else "nn", M, N, K, "PAIR")
        else:
            M, N, K = args[:3]
            key = ("tn", M, N, K, "W" + ("~" if args[12] is not None else ""))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = real(name, *args)
        e1.record()
        rec.append((key, e0, e1))
        return r

    ops.call = spy
    res = {n: collections.OrderedDict() for n in names}
    steps = {n: 0.0 for n in names}
    for rep in range(a.reps):
        for n in names:
            ops.gemm_tune.b16_bits = ops.gemm_tune.f32_bits = VARIANTS[n] if n in ('plan', 'no_p8', 'p8') or not f32 else 0
            rec.clear()
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record()
            step()
            t1.record()
            torch.cuda.synchronize()
            steps[n] += t0.elapsed_time(t1)
            for key, e0, e1 in rec:
                d = res[n].setdefault(key, [0, 0.0])
                d[0] += 1
                d[1] += e0.elapsed_time(e1) * 1e3
    ops.call = real
    ops.gemm_tune.b16_bits = ops.gemm_tune.f32_bits = 0
    base = res[names[0]]
    print(f"# {a.config}: in-situ us per call (mean over {a.reps} steps, with per-call events), calls per step, per-step total per variant")
    print("# step ms: " + "  ".join(f"{n} {steps[n] / a.reps:.2f}" for n in names))
    tot = {n: sum(v[1] for v in res[n].values()) / a.reps for n in names}
    print("# traced GEMM ms per step: " + "  ".join(f"{n} {tot[n] / 1e3:.3f}" for n in names))
    print(f"{'mode':4s} {'M':>6s} {'N':>6s} {'K':>6s} {'epi':8s} {'x':>3s} " + " ".join(f"{n:>9s}" for n in names) + "   d/step us")
    rows = sorted(base.items(), key=lambda kv: -kv[1][1])
    for key, v in rows:
        per = v[1] / v[0]
        if per < a.min_us:
            continue
        cnt = v[0] // a.reps
        cells = [res[n][key][1] / res[n][key][0] if key in res[n] else float("nan") for n in names]
        print(f"{key[0]:4s} {key[1]:6d} {key[2]:6d} {key[3]:6d} {key[4]:8s} {cnt:3d} " + " ".join(f"{c:9.1f}" for c in cells) + f"   {cnt * (cells[0] - min(cells)):8.1f}")


if __name__ == "__main__":
    main()
