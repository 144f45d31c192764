#!/usr/bin/env python3
"""In-situ per-call GEMM timing of ONE train step of a BASELINE config (--config, default Sub_GC_Kar) (packed decoder by default): every `ops.gemm`
call is bracketed by events on the stream it runs on; calls are grouped by (layout, M-bucket, N, K, epilogue).

    python tools/gemm_breakdown.py [--unpacked] [--batch 128] [--top 40]
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
import subgc.functions as F_  # noqa: E402
import subgc.functions_packed as FP  # noqa: E402
import subgc.models as models  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--unpacked", action="store_true")
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--top", type=int, default=45)
    ap.add_argument("--config", default="kar", choices=["kar", "full_gc_kar", "flickr"])
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(1234)
    if a.config == "kar":
        model = models.setup(argparse.Namespace(**bench.KAR)).to(dev).train()
        b = {k: v.to(dev) for k, v in synthetic.make_train_batch(a.batch, seed=0).items()}
    else:
        cfg = bench.CONFIGS[a.config]
        model = models.setup(argparse.Namespace(**cfg["opt"])).to(dev).train()
        b = {k: v.to(dev) for k, v in synthetic.make_train_batch(cfg["batch"], seed=1000, **cfg["data"]).items()}
    model.packed_decoder = not a.unpacked
    lw = models.LossWrapper(model, None)

    def step():
        model.flatten_grads()
        out = lw(b["fc_feats"], b["att_feats"], b["labels"], b["masks"], b["att_masks"], None, None, None, b["obj_dist"], None, b["rel_ind"],
                 None, b["pred_dist"], b["gpn_obj_ind"], b["gpn_pred_ind"], b["gpn_nrel_ind"], b["gpn_pool_mtx"])
        models.total_loss(out).backward()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    rec = []
    real = ops.gemm

    def timed(a_, b_, out, *, ta=False, tb=False, m_dev=None, a_rows=None, **kw):
        M = a_.size(1) if ta else a_.size(0)
        K = a_.size(0) if ta else a_.size(1)
        N = b_.size(0) if tb else b_.size(1)
        if a_rows is not None:
            M = a_rows.numel()
        if m_dev is not None:
            r = int(m_dev.item())
            M, K = (M, min(K, r)) if ta else (min(M, r), K)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        real(a_, b_, out, ta=ta, tb=tb, m_dev=m_dev, a_rows=a_rows, **kw)
        e1.record()
        epi = "".join(c for c, on in (("b", kw.get("bias") is not None), ("+", kw.get("add") is not None), ("r", kw.get("relu")),
                                      ("d", kw.get("keep") is not None), ("A", kw.get("accum"))) if on)
        rec.append((("tn" if ta else "nt" if tb else "nn"), M, N, K, epi, e0, e1))
        return out

    for mod in (ops, F_, FP):
        if getattr(mod, "gemm", None) is real:
            mod.gemm = timed
    ops.gemm = timed
    real_planes, real_lstm = ops.gemm_planes, ops.lstm_fwd_gemm

    def timed_planes(a_, b_, planes, *, ta=False, tb=False):
        M, K, N = a_.size(0), a_.size(1), b_.size(1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = real_planes(a_, b_, planes, ta=ta, tb=tb)
        e1.record()
        rec.append(("nn", M, N, K, "P%d" % r[0], e0, e1))
        return r

    def timed_lstm(x, w, *rest, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        real_lstm(x, w, *rest, **kw)
        e1.record()
        rec.append(("nt", x.size(0), w.size(0), x.size(1), "L", e0, e1))      # includes the cell kernel

    ops.gemm_planes, ops.lstm_fwd_gemm = timed_planes, timed_lstm
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    step()
    t1.record()
    torch.cuda.synchronize()
    agg = collections.OrderedDict()
    for mode, M, N, K, epi, e0, e1 in rec:
        mb = M if M > 1536 else (-(-M // 256) * 256 if M > 640 else 640 if M > 512 else 512 if M > 384 else 384 if M > 256 else 256 if M > 128 else 128)
        key = (mode, mb if mode != "tn" else M, N, K if mode != "tn" else (K // 1000) * 1000, epi)
        d = agg.setdefault(key, [0, 0.0, 0.0])
        d[0] += 1
        d[1] += e0.elapsed_time(e1) * 1e3
        d[2] += 2.0 * M * N * K
    tot_us = sum(v[1] for v in agg.values())
    tot_fl = sum(v[2] for v in agg.values())
    print(f"step {t0.elapsed_time(t1):.2f} ms (with per-call events); {len(rec)} gemm calls, {tot_us / 1e3:.2f} ms, {tot_fl / tot_us / 1e6:.1f} TF/s")
    print(f"{'mode':4s} {'M<=':>6s} {'N':>6s} {'K~':>6s} {'epi':4s} {'calls':>5s} {'us/call':>8s} {'total ms':>9s} {'%':>5s} {'TF/s':>6s}")
    for key, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[: a.top]:
        print(f"{key[0]:4s} {key[1]:6d} {key[2]:6d} {key[3]:6d} {key[4]:4s} {v[0]:5d} {v[1] / v[0]:8.1f} {v[1] / 1e3:9.3f} {100 * v[1] / tot_us:5.1f} {v[2] / v[1] / 1e6:6.1f}")


if __name__ == "__main__":
    main()
