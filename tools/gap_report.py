#!/usr/bin/env python3
"""Idle gaps of the GPU inside a rocprofv3 --kernel-trace (csv): sorts the dispatches by start time and reports, for the steady-state
part of the trace, total busy / idle time and the largest gaps with the kernels on both sides -- where a step waits for the host.

    python tools/gap_report.py <dir with *kernel_trace.csv> [min_gap_us=5] [skip_first_frac=0.5]"""
import csv, glob, re, sys, collections
d = sys.argv[1]
min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 5.0
skip = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    m = re.search(r"(\w+_kernel|__amd\w+|\w+Functor\w*|\w+)(?=[<(]|$)", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", ""))
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(1) if m else r["Kernel_Name"][:40]))
rows.sort()
rows = rows[int(len(rows) * skip):]
busy = idle = 0
gaps = []
end = rows[0][1]
for (s, e, k), prev in zip(rows[1:], rows[:-1]):
    g = s - end
    if g > 0:
        idle += g
        if g > min_gap * 1e3:
            gaps.append((g / 1e3, prev[2], k))
    end = max(end, e)
for s, e, k in rows:
    busy += e - s
span = rows[-1][1] - rows[0][0]
print(f"{len(rows)} dispatches over {span / 1e6:.2f} ms: kernel time {busy / 1e6:.2f} ms, idle {idle / 1e6:.2f} ms ({100 * idle / span:.1f} %), "
      f"{len(gaps)} gaps > {min_gap} us = {sum(g for g, _, _ in gaps) / 1e3:.2f} ms")
by = collections.defaultdict(lambda: [0, 0.0])
for g, a, b in gaps:
    by[(a, b)][0] += 1
    by[(a, b)][1] += g
for (a, b), (n, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{t:9.1f} us in {n:4d} gaps   after {a}   before {b}")
small = [max(0, s - pe) for (s, e, k), (ps, pe, pk) in zip(rows[1:], rows[:-1])]
print("gaps <= %g us: %d, total %.2f ms, mean %.2f us" % (min_gap, sum(1 for g in small if g <= min_gap * 1e3), sum(g for g in small if g <= min_gap * 1e3) / 1e6,
      sum(g for g in small if g <= min_gap * 1e3) / 1e3 / max(1, sum(1 for g in small if g <= min_gap * 1e3))))
