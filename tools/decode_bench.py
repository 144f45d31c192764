#!/usr/bin/env python3
"""Decode-only benchmark (greedy, one image per call like the reference's test.py)."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "sub-gc_amd"), ROOT]
import torch
import bench
import subgc.models as models

dev = torch.device("cuda:0")
torch.manual_seed(0)
m = models.setup(argparse.Namespace(**bench.KAR))
images = int(sys.argv[1]) if len(sys.argv) > 1 else 8
M = int(sys.argv[2]) if len(sys.argv) > 2 else 50
print(bench.decode_bench(m.state_dict(), dev, images, M))
