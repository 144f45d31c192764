#!/usr/bin/env python3
"""Where one reference-shaped greedy decode call (one image) spends its time: encoder launches, candidate scoring + NMS, the copies into the
graph's static buffers, the host read of the survivor count, the graph replay.  Two passes: host time per phase (asynchronous launches,
the GPU may lag) and, with a synchronize after every phase, the time each phase takes on the device when nothing overlaps.
    python tools/decode_phases.py [images] [M]"""
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
for b in batches:
    m(*synthetic.sample_args(b), opt=sopt, mode="sample")
torch.cuda.synchronize()

acc, SYNC = {}, [False]


def wrap(owner, name, label):
    real = getattr(owner, name)

    def f(*a, **k):
        t0 = time.perf_counter()
        r = real(*a, **k)
        if SYNC[0]:
            torch.cuda.synchronize()
        acc[label] = acc.get(label, 0.0) + time.perf_counter() - t0
        return r
    setattr(owner, name, f)


wrap(m, "_encode", "encode")
wrap(sampling, "score_candidates", "score+nms")
wrap(sampling._FrontBuffers, "load", "front copies")
wrap(sampling._GraphedLoop, "run", "replay+clones")
for sync in (False, True):
    SYNC[0] = sync
    acc.clear()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for b in batches:
        m(*synthetic.sample_args(b), opt=sopt, mode="sample")
    torch.cuda.synchronize()
    tot = 1e3 * (time.perf_counter() - t0) / images
    parts = ", ".join(f"{k} {1e3 * v / images:.3f}" for k, v in acc.items())
    print(f"sync_after_each_phase={sync}: {tot:.3f} ms/image; ms per phase: {parts}; rest (host read, result clones, python) {tot - 1e3 * sum(acc.values()) / images:.3f}", flush=True)
