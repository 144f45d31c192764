#!/usr/bin/env python3
"""Launches with KNOWN HBM byte counts for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950
(MI355X_MICROARCH.md "HBM": FETCH_SIZE halves wide coalesced reads; "calibrate on a known byte count in your own pattern").

Run under `rocprofv3 --pmc FETCH_SIZE --kernel-trace` and `--pmc WRITE_SIZE --kernel-trace` (tools/pmc_calibrate2.sh); the
report (tools/pmc_calibrate2_report.py) divides the counter by the bytes below.  Every buffer is larger than the 256 MiB
Infinity Cache or touched exactly once, so nothing can be served on-die.

  gemm 1 WG   : subgc_gemm_f32 NT, M = N = 64, K = 262144, ops.gemm_tune(no_splitk, no_skinny) -> ONE 64x64 workgroup (Grid_Size 256);
                each operand row is read exactly once: 2 x 64 x K x 4 B = 134,217,728 B
  gemm 4 WG   : M = N = 128 -> four 64x64 workgroups on four XCDs; every operand panel is read by TWO workgroups whose L2s
                are private: 4 x 134,217,728 = 536,870,912 B cross the fabric (268,435,456 B if the Infinity Cache dedups)
  sumsq       : subgc_sumsq_f32 over 2 GiB: float4 streaming read, 2,147,483,648 B, no write
  fill        : subgc_fill_f32 over 2 GiB: streaming write, 2,147,483,648 B, no read
  copy2d      : subgc_copy2d_f32 1 GiB -> 1 GiB: read 1,073,741,824 B and write 1,073,741,824 B
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sub-gc_amd"))
import torch  # noqa: E402

from subgc import ops  # noqa: E402

dev = "cuda:0"
K = 262144
g = torch.Generator().manual_seed(0)
for M in (64, 128):
    a = torch.randn(M, K, generator=g).to(dev)
    b = torch.randn(M, K, generator=g).to(dev)
    out = torch.empty(M, M, device=dev)
    big = torch.empty(1 << 28, device=dev).normal_()          # 1 GiB of other data through L2 / MALL between the launches
    torch.cuda.synchronize()
    with ops.gemm_tune(no_splitk=True, no_skinny=True):        # M = 64 must go to ONE tiled workgroup, not the split-K / weight-streaming forms
        ops.gemm(a, b, out, tb=True)
    torch.cuda.synchronize()
    del big
n = 1 << 29                                                    # 2 GiB of fp32
x = torch.empty(n, device=dev)
acc = torch.zeros(1, device=dev)
ops.fill_(x, 1.0)
torch.cuda.synchronize()
ops.sumsq(x, acc)
torch.cuda.synchronize()
src, dst = x[: n // 2].view(1 << 14, -1), torch.empty(n // 2, device=dev).view(1 << 14, -1)
ops.copy2d(src, dst)
torch.cuda.synchronize()
print("ok", float(acc))
