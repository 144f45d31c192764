#!/usr/bin/env python3
"""Batched decode only (sample_images, `group` images per decode batch): wall time per phase, for rocprofv3 runs.

    python tools/decode_batched_bench.py [images=128] [group=64] [M=50]
"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "sub-gc_amd"), ROOT]
import torch
import bench
import subgc.models as models
from subgc import synthetic
from subgc.models import sampling

dev = torch.device("cuda:0")
torch.manual_seed(0)
images = int(sys.argv[1]) if len(sys.argv) > 1 else 128
group = int(sys.argv[2]) if len(sys.argv) > 2 else 64
M = int(sys.argv[3]) if len(sys.argv) > 3 else 50
opt = argparse.Namespace(**dict(bench.KAR, test_LSTM=1, gpn_nms_thres=0.75, gpn_max_subg=10))
m = models.setup(opt).to(dev).eval()
batches = [{k: v.to(dev) for k, v in synthetic.make_test_batch(M, seed=500 + i).items()} for i in range(images)]
sopt = dict(sample_max=1, beam_size=1)
m.sample_images(batches[:group], opt=sopt)
torch.cuda.synchronize()


def phases(ims):
    t = [time.perf_counter()]
    att = torch.cat([im["att_feats"][:1] for im in ims])
    I, N, _ = att.shape
    X2 = m._encode(att, torch.cat([im["obj_dist"][:1] for im in ims]), torch.cat([im["pred_dist"][:1] for im in ims]),
                   torch.cat([im["rel_ind"][:1] for im in ims])).reshape(I * N, m.GCN_dim).contiguous()
    torch.cuda.synchronize(); t.append(time.perf_counter())
    rows = [(i, im["gpn_obj_ind"], im["att_masks"], im["gpn_pool_mtx"]) for i, im in enumerate(ims)]
    sel = sampling.select_subgraphs(m, X2, N, rows)
    torch.cuda.synchronize(); t.append(time.perf_counter())
    out = sampling.decode(m, X2, N, sel, sopt)
    torch.cuda.synchronize(); t.append(time.perf_counter())
    return out, [1e3 * (b - a) for a, b in zip(t, t[1:])]


with torch.no_grad():
    acc = [0.0, 0.0, 0.0]
    for i in range(0, images, group):
        _, ms = phases(batches[i:i + group])
        acc = [a + b for a, b in zip(acc, ms)]
print("phase ms per group (encode, select, decode):", [round(a / (images / group), 3) for a in acc])
t0 = time.perf_counter()
tokens = 0
for i in range(0, images, group):
    for r in m.sample_images(batches[i:i + group], opt=sopt):
        tokens += r[0].size(0) * r[0].size(1)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print({"decode_batched_tokens_per_s": round(tokens / dt, 1), "ms_per_group": round(1e3 * dt / (images / group), 3), "tokens": tokens})
