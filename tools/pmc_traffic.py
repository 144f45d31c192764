#!/usr/bin/env python3
"""HBM traffic of the GEMM kernel family per `subgc_gemm_f32` launch, from two rocprofv3 PMC passes.

    python tools/pmc_traffic.py <FETCH_SIZE counter_collection.csv> <WRITE_SIZE counter_collection.csv> \
           --steps-total 6 --launches-per-step 142 --alg-bytes-per-launch 5.1e7 > profiles/rNN_pmc_traffic.json

Collected as MI355X_MICROARCH.md "HBM" prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE `--pmc` passes of the
same bench command (tools/pmc_traffic.sh); rocprofv3 reports both in KiB.  The guide warns that on gfx950 FETCH_SIZE
counts HALF the bytes of wide coalesced reads and asks for a calibration on known byte counts in the kernel's own access
pattern.  Round 2 redid it (tools/pmc_calibrate2.sh -> profiles/r02_pmc_calibration.txt): ONE 64x64 workgroup of this GEMM
(Grid_Size 256) reading 134,217,728 B counts 67,120,448 B; four workgroups reading 536,870,912 B through four private
L2s count 268,478,336 B; a 2 GiB float4 streaming read counts 1 GiB -- FETCH_SIZE x 2.0 in every case -- while WRITE_SIZE
is exact (2 GiB fill: 2,147,483,648 B; the GEMM's 16 / 64 KB of results to the byte).  (Round 1's "factor 1.0" compared the
four-workgroup launch with the bytes of ONE pass over the operands; each panel is fetched by two XCDs there.)  A logical launch = one `subgc_gemm_f32` call = its main kernel
plus, in split-K form, the reduce kernel (both are inside the HIP-event bracket that times it in bench.py).
"""
import argparse
import collections
import csv
import json

FETCH_FACTOR = 2.0          # profiles/r02_pmc_calibration.txt
FAMILY = ("gemm_f32_kernel", "gemm_f32_splitk_kernel", "splitk_reduce_kernel", "gemm_skinny",
          "gemm_bf16_kernel", "gemm_bf16_splitk_kernel", "gemm_bf16_p8_kernel", "gemm_bf16_p8_splitk_kernel",
          "splitk_reduce_b16_kernel")                                                          # bf16 configs: both GEMM families run
REDUCE = ("splitk_reduce_kernel", "splitk_reduce_b16_kernel")
STEP_MARK = "live_plan_kernel"      # launched exactly once per (packed) train step: counts the profiled steps when one call may be two kernels


def totals(path, counter):
    tot = collections.defaultdict(float)
    disp = collections.defaultdict(set)
    marks = set()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"]
        if STEP_MARK in name:
            marks.add(r["Dispatch_Id"])
        fam = next((f for f in FAMILY if f in name), None)
        if fam is None:
            continue
        tot[fam] += float(r["Counter_Value"])
        disp[fam].add(r["Dispatch_Id"])
    return tot, {k: len(v) for k, v in disp.items()}, len(marks)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_csv")
    ap.add_argument("write_csv")
    ap.add_argument("--steps-total", type=int, default=0, help="steps of the profiled command (default: inferred from the dispatch count)")
    ap.add_argument("--launches-per-step", type=int, required=True, help="subgc_gemm_f32 calls per step (bench.py roofline.launches_per_step)")
    ap.add_argument("--alg-bytes-per-launch", type=float, default=None)
    a = ap.parse_args()
    f, fd, steps_seen = totals(a.fetch_csv, "FETCH_SIZE")
    w, wd, _ = totals(a.write_csv, "WRITE_SIZE")
    fetch = FETCH_FACTOR * 1024.0 * sum(f.values())  # KiB -> B, x the calibrated factor (see above)
    write = 1024.0 * sum(w.values())
    mains = sum(v for k, v in fd.items() if k not in REDUCE)                      # one main kernel per subgc_gemm_f32 / _bf16 call
    if not a.steps_total:
        if mains % a.launches_per_step == 0:
            a.steps_total = mains // a.launches_per_step
        elif steps_seen and steps_seen * a.launches_per_step <= mains < 2 * steps_seen * a.launches_per_step:
            # a call of the bf16 GEMM may be TWO main kernels (row cut: whole rounds of tiles + the remaining rows, gemm_bf16.hip run()):
            # the steps are counted on a once-per-step kernel instead, the bytes of both kernels belong to the one logical launch
            a.steps_total = steps_seen
        else:
            raise SystemExit(f"{mains} GEMM dispatches are not a multiple of {a.launches_per_step} launches per step ({steps_seen} steps seen)")
    elif mains < a.steps_total * a.launches_per_step:
        raise SystemExit(f"{mains} GEMM dispatches < {a.steps_total} steps x {a.launches_per_step} launches")
    launches = a.steps_total * a.launches_per_step
    out = {
        "counters": "FETCH_SIZE x 2.0 (gfx950 halving, calibrated: profiles/r02_pmc_calibration.txt) + WRITE_SIZE (exact), separate --pmc passes",
        "fetch_factor": FETCH_FACTOR,
        "steps_profiled": a.steps_total, "gemm_launches_per_step": a.launches_per_step,
        "kernel_dispatches": fd, "main_kernels_per_launch": mains / (a.steps_total * a.launches_per_step),
        "fetch_bytes_per_step": fetch / a.steps_total, "write_bytes_per_step": write / a.steps_total,
        "traffic_bytes_per_launch": (fetch + write) / launches,
        "per_kernel_bytes_per_dispatch": {k: {"fetch": FETCH_FACTOR * 1024.0 * f[k] / fd[k], "write": 1024.0 * w.get(k, 0.0) / max(wd.get(k, 1), 1)} for k in f},
    }
    if a.alg_bytes_per_launch:
        out["algorithmic_bytes_per_launch"] = a.alg_bytes_per_launch
        out["traffic_over_algorithmic"] = out["traffic_bytes_per_launch"] / a.alg_bytes_per_launch
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
