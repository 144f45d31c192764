#!/usr/bin/env python3
"""Which package lines issue the plumbing launches of one bf16 train step (casts, fills, add_n, copies, separate column sums): every such
`ops.call` of one real step of Full_GC_Kar and the Flickr shape with its two innermost package frames -- the list behind "what is left to
merge" in DESIGN section 8."""
import os, sys, argparse, collections, traceback, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "sub-gc_amd")); sys.path.insert(0, R)
import bench
import subgc.models as models
from subgc import synthetic, ops
DEV = "cuda:0"
for cfgname in (sys.argv[1:] or ("full_gc_kar", "flickr")):
    cfg = bench.CONFIGS[cfgname] if cfgname != "kar" else {"opt": bench.KAR, "batch": 128, "data": {}}
    torch.manual_seed(5)
    m = models.setup(argparse.Namespace(**cfg["opt"])).to(DEV).train()
    lw = models.LossWrapper(m, None)
    b = {k: v.to(DEV) for k, v in synthetic.make_train_batch(cfg["batch"], seed=3, **cfg["data"]).items()}
    def step():
        m.flatten_grads()
        models.total_loss(lw(*bench.lw_args(b))).backward()
    step(); step()
    seen = collections.Counter()
    real = ops.call
    def spy(name, *a):
        if name in ("subgc_cast_f32_bf16", "subgc_fill_f32", "subgc_add_n_f32", "subgc_copy2d_b16", "subgc_colsum_bf16", "subgc_colsum_f32", "subgc_transpose_f32_bf16",
                    "subgc_copy2d_f32", "subgc_gather_rows_f32", "subgc_scatter_add_rows_f32", "subgc_fill2d_f32", "subgc_relu_bwd"):
            st = [f for f in traceback.extract_stack() if "subgc" in f.filename and "ops.py" not in f.filename and "_casts" not in f.filename]
            where = " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in st[-2:][::-1])
            shape = (a[4], a[5]) if name == "subgc_cast_f32_bf16" else ""
            seen[(name, where, str(shape))] += 1
        return real(name, *a)
    ops.call = spy
    step()
    ops.call = real
    print("==", cfgname)
    for (n, w, s), c in sorted(seen.items(), key=lambda kv: (kv[0][0], -kv[1])):
        print(f"{c:3d} {n:28s} {s:16s} {w}")
