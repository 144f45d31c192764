#!/usr/bin/env python3
"""counter_collection.csv of tools/pmc_calibrate2.py -> counted bytes / known bytes per calibration launch."""
import collections
import csv
import sys

KNOWN = [  # (label, kernel-name substring, Grid_Size filter or None, known read bytes, known write bytes)
    ("gemm 1 WG (64x64, K=262144)", "gemm_f32_kernel", 256, 2 * 64 * 262144 * 4, 64 * 64 * 4),
    ("gemm 4 WG (128x128 as 4 x 64x64)", "gemm_f32_kernel", 1024, 4 * 2 * 64 * 262144 * 4, 128 * 128 * 4),
    ("sumsq 2 GiB (float4 streaming read)", "sumsq", None, 1 << 31, 0),
    ("fill 2 GiB (streaming write)", "fill_kernel", None, 0, 1 << 31),
    ("copy2d 1 GiB -> 1 GiB", "copy2d_kernel", None, 1 << 30, 1 << 30),
]
path, counter = sys.argv[1], sys.argv[2]
rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
print(f"# {counter} (rocprofv3 reports KiB) from {path.split('/')[-1]}")
for label, sub, grid, rd, wr in KNOWN:
    per = collections.defaultdict(float)
    for r in rows:
        if sub in r["Kernel_Name"] and (grid is None or int(r["Grid_Size"]) == grid):
            per[r["Dispatch_Id"]] += float(r["Counter_Value"])
    if not per:
        print(f"{label:40s} no dispatch found")
        continue
    vals = sorted(per.values())
    counted = vals[-1] * 1024.0                                 # the fill kernel also runs for small scratch: take the big launch
    known = rd if counter == "FETCH_SIZE" else wr
    ratio = counted / known if known else float("nan")
    print(f"{label:40s} dispatches {len(per):3d}  counted {counted:16,.0f} B  known {known:16,d} B  counted/known {ratio:6.3f}")
