#!/usr/bin/env python3
"""Informational timings of the OTHER BASELINE.json configs (they are parity-test cases, not bench lines):
config 3  Full_GC_Kar (4 GCN layers + BatchNorm, no sGPN, attention over all 36 nodes), batch 256, bf16 GEMM mode;
config 5  Flickr stress shape (N=101, K=301, D=4096, L=2048, V+1=7001), 64 images/GPU, bf16 GEMM mode;
each also in the fp32 mode.  Train fwd+bwd through LossWrapper on a resident synthetic batch.

    python tools/other_configs_bench.py [--steps 8]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "sub-gc_amd"), ROOT]
import torch  # noqa: E402

import bench  # noqa: E402
from subgc import ops, synthetic  # noqa: E402
import subgc.models as models  # noqa: E402

CONFIGS = {
    "full_gc_kar_b256": (dict(use_gpn=0, noun_fuse=0, pred_emb_type=2, gcn_layers=4, gcn_residual=1, gcn_bn=1), dict(), 256),
    "flickr_stress_b64": (dict(vocab_size=7000, fc_feat_size=4096, att_feat_size=4096, gcn_dim=2048),
                          dict(N=101, K=301, D=4096, n_edges=300, max_nodes=30, vocab=7000), 64),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=8)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    out = {}
    for name, (over, data, B) in CONFIGS.items():
        torch.manual_seed(1)
        m = models.setup(argparse.Namespace(**dict(bench.KAR, **over))).to(dev).train()
        lw = models.LossWrapper(m, None)
        b = {k: v.to(dev) for k, v in synthetic.make_train_batch(B, seed=5, **data).items()}

        def step():
            m.flatten_grads()
            o = lw(*bench.lw_args(b))
            loss = o["lang_loss"] + (o["gpn_loss"] if o["gpn_loss"] is not None else 0.0)
            loss.backward()
            return loss

        for mode in ("f32", "bf16"):
            with ops.gemm_mode(mode):
                for _ in range(3):
                    step()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(a.steps):
                    loss = step()
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
            out[f"{name}/{mode}"] = {"images_per_s": round(B * a.steps / dt, 1), "ms_per_step": round(1e3 * dt / a.steps, 2), "loss": round(float(loss), 4)}
        del m, lw, b
        torch.cuda.empty_cache()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
