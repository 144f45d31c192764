#!/usr/bin/env python3
"""One decode leg of bench.py as a stand-alone workload for rocprofv3 (kernel trace / PMC passes):

    python tools/decode_legs.py greedy_batched|beam2_batched|mrnn [passes=3]

Same models, seeds and shapes as bench.decode_bench / bench.mrnn_decode_leg: one warm-up pass, then `passes` passes."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "sub-gc_amd"), ROOT]
import torch
import bench
import subgc.models as models
from subgc import synthetic

leg = sys.argv[1]
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda:0")
torch.manual_seed(1234)
if leg == "mrnn":
    torch.manual_seed(0)
    m = models.setup(argparse.Namespace(**dict(bench.KAR, **bench.MRNN))).to(dev).eval()
    batches = [{k: v.to(dev) for k, v in synthetic.make_test_batch(500, seed=900 + i).items()} for i in range(6)]
    sopt = dict(sample_max=1, beam_size=1)
    fn = lambda: [m(*synthetic.sample_args(b), opt=sopt, mode="sample") for b in batches]
else:
    m = models.setup(argparse.Namespace(**dict(bench.KAR, test_LSTM=1, gpn_nms_thres=0.75, gpn_max_subg=10))).to(dev).eval()
    batches = [{k: v.to(dev) for k, v in synthetic.make_test_batch(50, seed=500 + i).items()} for i in range(256)]
    sopt = dict(sample_max=1, beam_size=2 if leg == "beam2_batched" else 1)
    fn = lambda: m.sample_images(batches, opt=sopt)
from subgc import _lib
with torch.no_grad():
    fn()
    torch.cuda.synchronize()
    _lib.prof_enable("gemm")                       # counts the GEMM CALLS (a call may be two main kernels: row cut, gemm_*.hip)
    for _ in range(passes):
        fn()
    torch.cuda.synchronize()
    calls = _lib.prof_collect("gemm")[0]
    _lib.prof_enable("gemm", False)
print(leg, "done", passes, "passes")
print("gemm_calls_per_pass", calls // passes)
