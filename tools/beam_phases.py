import argparse, os, sys, time
sys.path[:0] = ["sub-gc_amd", "."]
import torch, bench
import subgc.models as models
from subgc import synthetic, beam
from subgc.models import sampling
dev = torch.device("cuda:0"); torch.manual_seed(0)
opt = argparse.Namespace(**dict(bench.KAR, test_LSTM=1, gpn_nms_thres=0.75, gpn_max_subg=10))
m = models.setup(opt).to(dev).eval()
batches = [{k: v.to(dev) for k, v in synthetic.make_test_batch(50, seed=500 + i).items()} for i in range(32)]
sopt = dict(sample_max=1, beam_size=2)
for b in batches: m(*synthetic.sample_args(b), opt=sopt, mode="sample")
torch.cuda.synchronize()
acc = {}
def wrap(owner, name, label, sync):
    real = getattr(owner, name)
    def f(*a, **k):
        t0 = time.perf_counter(); r = real(*a, **k)
        if sync: torch.cuda.synchronize()
        acc[label] = acc.get(label, 0.0) + time.perf_counter() - t0
        return r
    setattr(owner, name, f)
wrap(m, "_encode", "encode", True); wrap(sampling, "score_candidates", "score+nms", True); wrap(sampling._FrontBuffers, "load", "front copies", True)
wrap(beam.DeviceSearch, "collect", "collect (host)", False)
real_run = sampling._GraphedBeam.run
def run(self, pr):
    if pr is not None: sampling._load_prepared(self.pr, pr)
    t0 = time.perf_counter(); self.graph.replay(); torch.cuda.synchronize(); acc["replay (sync)"] = acc.get("replay (sync)", 0.0) + time.perf_counter() - t0
    return self.ds.collect()
sampling._GraphedBeam.run = run
t0 = time.perf_counter()
for b in batches: m(*synthetic.sample_args(b), opt=sopt, mode="sample")
torch.cuda.synchronize()
tot = 1e3 * (time.perf_counter() - t0) / 32
print(f"{tot:.3f} ms/image (phases synchronised):", {k: round(1e3 * v / 32, 3) for k, v in acc.items()})
