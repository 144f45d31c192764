#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace --stats output directory into a small text summary
(per-kernel calls / total / average / share), the file that gets committed under profiles/."""
import csv
import glob
import os
import sys


def main(d, out):
    files = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
    if not files:
        files = glob.glob(os.path.join(d, "**", "*stats*.csv"), recursive=True)
    rows = []
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append(r)
    if not rows:
        print("no stats csv under", d); return 1
    keys = rows[0].keys()
    name = next(k for k in keys if k.lower() in ("name", "kernelname", "kernel_name"))
    calls = next(k for k in keys if "calls" in k.lower())
    tot = next(k for k in keys if "total" in k.lower() and "dur" in k.lower())
    avg = next(k for k in keys if "average" in k.lower())
    agg = {}
    for r in rows:
        a = agg.setdefault(r[name], [0, 0.0])
        a[0] += int(r[calls]); a[1] += float(r[tot])
    total = sum(v[1] for v in agg.values())
    # With the recurrence cut into two chains on two streams, kernels overlap: the SUM of durations counts that time twice.  From the
    # kernel trace (same run): the union of the dispatch intervals, for all kernels and for the GEMM family (gemm_* + splitk_reduce*).
    union_lines = []
    traces = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    if traces:
        iv_all, iv_gemm = [], []
        with open(traces[0]) as fh:
            for r in csv.DictReader(fh):
                s_, e_ = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
                iv_all.append((s_, e_))
                if "gemm_" in r["Kernel_Name"] or "splitk_reduce" in r["Kernel_Name"]:
                    iv_gemm.append((s_, e_))

        def union(iv):
            iv.sort()
            busy, lo, hi = 0, None, None
            for a, b in iv:
                if lo is None:
                    lo, hi = a, b
                elif a <= hi:
                    hi = max(hi, b)
                else:
                    busy += hi - lo
                    lo, hi = a, b
            return busy + (hi - lo if lo is not None else 0)
        for tag, iv in (("all kernels", iv_all), ("GEMM family (gemm_* + splitk_reduce*)", iv_gemm)):
            if iv:
                sm = sum(b - a for a, b in iv)
                union_lines.append(f"# {tag}: sum of durations {sm/1e6:.3f} ms, union of intervals {union(list(iv))/1e6:.3f} ms (overlap x{sm/max(union(list(iv)),1):.3f})\n")
    with open(out, "w") as fh:
        fh.write(f"# rocprofv3 --kernel-trace --stats summary ({os.path.basename(d)}); durations in ns\n")
        fh.write(f"# total kernel time {total/1e6:.3f} ms over {sum(v[0] for v in agg.values())} launches\n")
        for ln in union_lines:
            fh.write(ln)
        fh.write(f"{'calls':>8} {'total_ms':>12} {'avg_us':>10} {'share':>7}  kernel\n")
        for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            fh.write(f"{c:8d} {t/1e6:12.3f} {t/c/1e3:10.2f} {100*t/total:6.2f}%  {k[:160]}\n")
    print(open(out).read())
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1], sys.argv[2]))
