import argparse, os, sys, time, cProfile, pstats, io
ROOT = "/root/repo"
sys.path[:0] = [os.path.join(ROOT, "sub-gc_amd"), ROOT]
import torch
import bench
from subgc import synthetic
import subgc.models as models
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = models.setup(argparse.Namespace(**bench.KAR)).to(dev).train()
lw = models.LossWrapper(m, None)
b = {k: v.to(dev) for k, v in synthetic.make_train_batch(4, seed=0).items()}
def step():
    m.flatten_grads()
    out = lw(*bench.lw_args(b))
    (out["lang_loss"] + out["gpn_loss"]).backward()
for _ in range(3): step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    step()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:6000])
