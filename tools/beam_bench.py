#!/usr/bin/env python3
"""Beam-search decode as test.sh runs it for Sub_GC_Kar (beam 2, NMS 0.75, <= 10 sub-graphs) and Full-GC-like beam 3:
one image per call and `group` images per decode batch.    python tools/beam_bench.py [images=32] [beam=2] [group=64]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "sub-gc_amd"), ROOT]
import torch
import bench
import subgc.models as models
from subgc import synthetic

dev = torch.device("cuda:0")
torch.manual_seed(0)
images = int(sys.argv[1]) if len(sys.argv) > 1 else 32
beam = int(sys.argv[2]) if len(sys.argv) > 2 else 2
group = int(sys.argv[3]) if len(sys.argv) > 3 else 64
opt = argparse.Namespace(**dict(bench.KAR, test_LSTM=1, gpn_nms_thres=0.75, gpn_max_subg=10))
m = models.setup(opt).to(dev).eval()
batches = [{k: v.to(dev) for k, v in synthetic.make_test_batch(50, seed=500 + i).items()} for i in range(images)]
sopt = dict(sample_max=1, beam_size=beam)
for b in batches[:2]:
    m(*synthetic.sample_args(b), opt=sopt, mode="sample")
torch.cuda.synchronize()
t0 = time.perf_counter()
tokens = 0
for b in batches:
    seq = m(*synthetic.sample_args(b), opt=sopt, mode="sample")[0]
    tokens += seq.numel()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print({"beam": beam, "one_image_ms": round(1e3 * dt / images, 3), "tokens_per_s": round(tokens / dt, 1)})
m.sample_images(batches[:group], opt=sopt)
torch.cuda.synchronize()
t0 = time.perf_counter()
tokens = 0
for i in range(0, images, group):
    for r in m.sample_images(batches[i:i + group], opt=sopt):
        tokens += r[0].numel()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print({"beam": beam, "group": group, "batched_ms_per_image": round(1e3 * dt / images, 3), "tokens_per_s": round(tokens / dt, 1)})
