#!/usr/bin/env python3
"""Decode as test.sh runs Sub_GC_S_MRNN: up to 1000 candidate sub-graphs per image, NMS 0.55, keep up to 1000, top-k
sampling (k = 3, T = 0.6), one image per call.    python tools/mrnn_decode_bench.py [images=8] [M=500]"""
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
images = int(sys.argv[1]) if len(sys.argv) > 1 else 8
M = int(sys.argv[2]) if len(sys.argv) > 2 else 500
opt = argparse.Namespace(**dict(bench.KAR, test_LSTM=1, gpn_nms_thres=0.55, gpn_max_subg=1000, use_topk_sampling=1, topk_temp=0.6, the_k=3))
m = models.setup(opt).to(dev).eval()
batches = [{k: v.to(dev) for k, v in synthetic.make_test_batch(M, seed=900 + i).items()} for i in range(images)]
sopt = dict(sample_max=1, beam_size=1)
for b in batches[:2]:
    m(*synthetic.sample_args(b), opt=sopt, mode="sample")
torch.cuda.synchronize()
t0 = time.perf_counter()
rows = tokens = 0
for b in batches:
    seq = m(*synthetic.sample_args(b), opt=sopt, mode="sample")[0]
    rows += seq.size(0); tokens += seq.numel()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print({"candidates": 2 * M, "kept_per_image": round(rows / images, 1), "ms_per_image": round(1e3 * dt / images, 2), "tokens_per_s": round(tokens / dt, 1)})
acc = [0.0, 0.0, 0.0]
with torch.no_grad():
    for im in batches:
        t = [time.perf_counter()]
        N = im["att_feats"].size(1)
        X2 = m._encode(im["att_feats"][:1], im["obj_dist"][:1], im["pred_dist"][:1], im["rel_ind"][:1]).reshape(N, m.GCN_dim).contiguous()
        torch.cuda.synchronize(); t.append(time.perf_counter())
        sel = sampling.select_subgraphs(m, X2, N, [(0, im["gpn_obj_ind"], im["att_masks"], im["gpn_pool_mtx"])])
        torch.cuda.synchronize(); t.append(time.perf_counter())
        sampling.decode(m, X2, N, sel, sopt)
        torch.cuda.synchronize(); t.append(time.perf_counter())
        acc = [a + 1e3 * (y - x) for a, x, y in zip(acc, t, t[1:])]
print("ms per image (encode, select incl. NMS, decode):", [round(a / images, 3) for a in acc])
