#!/usr/bin/env python3
"""HBM traffic of the GEMM kernel family per `subgc_gemm_f32` launch, from two rocprofv3 PMC passes.

    python tools/pmc_traffic.py <FETCH_SIZE counter_collection.csv> <WRITE_SIZE counter_collection.csv> \
           --steps-total 6 --launches-per-step 142 --alg-bytes-per-launch 5.1e7 > profiles/rNN_pmc_traffic.json

Collected as MI355X_MICROARCH.md "HBM" prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE `--pmc` passes of the
same bench command (tools/pmc_traffic.sh); rocprofv3 reports both in KiB.  The guide warns that on gfx950 FETCH_SIZE
can count HALF the bytes of wide streaming reads and asks for a calibration on the kernel's own access pattern:
tools/pmc_calibrate.sh runs one 128x128 tile over K = 262144 (each operand read exactly once: 268,435,456 B) and
FETCH_SIZE reports 262,181..262,186 KiB = 268.47 MB for the NT, NN and TN forms alike -- for this kernel's loads
(global_load_dwordx4 in 128-B row segments) the counter is exact, so NO correction factor is applied.
WRITE_SIZE is uncalibrated and taken as reported.  A logical launch = one `subgc_gemm_f32` call = its main kernel
plus, in split-K form, the reduce kernel (both are inside the HIP-event bracket that times it in bench.py).
"""
import argparse
import collections
import csv
import json

FAMILY = ("gemm_f32_kernel", "gemm_f32_splitk_kernel", "splitk_reduce_kernel", "gemm_skinny")


def totals(path, counter):
    tot = collections.defaultdict(float)
    disp = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"]
        fam = next((f for f in FAMILY if f in name), None)
        if fam is None:
            continue
        tot[fam] += float(r["Counter_Value"])
        disp[fam].add(r["Dispatch_Id"])
    return tot, {k: len(v) for k, v in disp.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_csv")
    ap.add_argument("write_csv")
    ap.add_argument("--steps-total", type=int, default=0, help="steps of the profiled command (default: inferred from the dispatch count)")
    ap.add_argument("--launches-per-step", type=int, required=True, help="subgc_gemm_f32 calls per step (bench.py roofline.launches_per_step)")
    ap.add_argument("--alg-bytes-per-launch", type=float, default=None)
    a = ap.parse_args()
    f, fd = totals(a.fetch_csv, "FETCH_SIZE")
    w, wd = totals(a.write_csv, "WRITE_SIZE")
    fetch = 1024.0 * sum(f.values())                # KiB -> B; calibrated factor 1.0 (see above)
    write = 1024.0 * sum(w.values())
    mains = sum(v for k, v in fd.items() if k != "splitk_reduce_kernel")          # one main kernel per subgc_gemm_f32 call
    if not a.steps_total:
        if mains % a.launches_per_step:
            raise SystemExit(f"{mains} GEMM dispatches are not a multiple of {a.launches_per_step} launches per step")
        a.steps_total = mains // a.launches_per_step
    elif mains != a.steps_total * a.launches_per_step:
        raise SystemExit(f"{mains} GEMM dispatches != {a.steps_total} steps x {a.launches_per_step} launches")
    launches = a.steps_total * a.launches_per_step
    out = {
        "counters": "FETCH_SIZE (calibrated on this kernel: factor 1.0) + WRITE_SIZE, separate --pmc passes, KiB",
        "steps_profiled": a.steps_total, "gemm_launches_per_step": a.launches_per_step,
        "kernel_dispatches": fd,
        "fetch_bytes_per_step": fetch / a.steps_total, "write_bytes_per_step": write / a.steps_total,
        "traffic_bytes_per_launch": (fetch + write) / launches,
        "per_kernel_bytes_per_dispatch": {k: {"fetch": 1024.0 * f[k] / fd[k], "write": 1024.0 * w.get(k, 0.0) / max(wd.get(k, 1), 1)} for k in f},
    }
    if a.alg_bytes_per_launch:
        out["algorithmic_bytes_per_launch"] = a.alg_bytes_per_launch
        out["traffic_over_algorithmic"] = out["traffic_bytes_per_launch"] / a.alg_bytes_per_launch
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
