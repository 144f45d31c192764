#!/usr/bin/env python3
"""How much of a rocprofv3 --kernel-trace runs with two or more kernels in flight (side-stream overlap), and which kernel pairs overlap.
    python tools/overlap_report.py <dir with *kernel_trace.csv> [skip_first_frac=0.5]"""
import csv, glob, re, sys, collections
d = sys.argv[1]
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"(\w+)", n)
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(1) if m else n[:30], r.get("Queue_Id", "")))
rows.sort()
rows = rows[int(len(rows) * skip):]
ev = []
for i, (s, e, k, q) in enumerate(rows):
    ev.append((s, 1, i)); ev.append((e, -1, i))
ev.sort()
active = set(); last = ev[0][0]; t1 = t2 = 0
pairs = collections.Counter()
for t, kind, i in ev:
    dt = t - last
    if len(active) == 1: t1 += dt
    elif len(active) >= 2:
        t2 += dt
        names = sorted(rows[j][2] for j in active)[:2]
        pairs[tuple(names)] += dt
    last = t
    if kind == 1: active.add(i)
    else: active.discard(i)
span = rows[-1][1] - rows[0][0]
print(f"{len(rows)} dispatches over {span/1e6:.2f} ms: one kernel in flight {t1/1e6:.2f} ms, two or more {t2/1e6:.2f} ms ({100*t2/span:.1f} %), queues: {sorted(set(r[3] for r in rows))}")
for (a, b), t in pairs.most_common(12):
    print(f"{t/1e3:9.1f} us  {a}  ||  {b}")
