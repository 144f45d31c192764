#!/usr/bin/env python3
"""How long does the host need to ENQUEUE one train step (no device sync inside the loop)?"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "sub-gc_amd"), ROOT]
import torch
import bench
from subgc import synthetic
import subgc.models as models

dev = torch.device("cuda:0")
torch.manual_seed(0)
m = models.setup(argparse.Namespace(**bench.KAR)).to(dev).train()
lw = models.LossWrapper(m, None)
b = {k: v.to(dev) for k, v in synthetic.make_train_batch(int(sys.argv[1]) if len(sys.argv) > 1 else 128, seed=0).items()}
def step():
    m.flatten_grads()
    out = lw(*bench.lw_args(b))
    (out["lang_loss"] + out["gpn_loss"]).backward()
for _ in range(3): step()
torch.cuda.synchronize()
for tag in ("enqueue-only", "with-sync"):
    t0 = time.perf_counter()
    for _ in range(5):
        step()
        if tag == "with-sync": torch.cuda.synchronize()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{tag}: host {1e3*(t1-t0)/5:.2f} ms/step, incl. drain {1e3*(t2-t0)/5:.2f} ms/step")
