#!/usr/bin/env python3
"""One-image decode (the reference's call shape): wall time per phase, synchronising between phases."""
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
images = int(sys.argv[1]) if len(sys.argv) > 1 else 32
M = int(sys.argv[2]) if len(sys.argv) > 2 else 50
opt = argparse.Namespace(**dict(bench.KAR, test_LSTM=1, gpn_nms_thres=0.75, gpn_max_subg=10))
m = models.setup(opt).to(dev).eval()
batches = [{k: v.to(dev) for k, v in synthetic.make_test_batch(M, seed=500 + i).items()} for i in range(images)]
sopt = dict(sample_max=1, beam_size=1)
for b in batches[:3]:
    m(*synthetic.sample_args(b), opt=sopt, mode="sample")
torch.cuda.synchronize()
acc = [0.0] * 4
with torch.no_grad():
    for im in batches:
        t = [time.perf_counter()]
        att = im["att_feats"][:1]
        N = att.size(1)
        X2 = m._encode(att, im["obj_dist"][:1], im["pred_dist"][:1], im["rel_ind"][:1]).reshape(N, m.GCN_dim).contiguous()
        torch.cuda.synchronize(); t.append(time.perf_counter())
        sel = sampling.select_subgraphs(m, X2, N, [(0, im["gpn_obj_ind"], im["att_masks"], im["gpn_pool_mtx"])])
        torch.cuda.synchronize(); t.append(time.perf_counter())
        out = sampling.decode(m, X2, N, sel, sopt)
        torch.cuda.synchronize(); t.append(time.perf_counter())
        g = m._graph_cache[next(iter(m._graph_cache))] if getattr(m, "_graph_cache", None) else None
        if g is not None:
            g.graph.replay()
        torch.cuda.synchronize(); t.append(time.perf_counter())
        acc = [a + 1e3 * (y - x) for a, x, y in zip(acc, t, t[1:])]
print("ms per image (encode, select, decode incl. prepare, bare graph replay):", [round(a / images, 3) for a in acc])
t0 = time.perf_counter()
for b in batches:
    m(*synthetic.sample_args(b), opt=sopt, mode="sample")
torch.cuda.synchronize()
print("end-to-end ms per image:", round(1e3 * (time.perf_counter() - t0) / images, 3))
