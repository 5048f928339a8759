#!/usr/bin/env python3
"""How many kernels run at once, over the second half of a rocprofv3 --kernel-trace run (the timed steps of bench.py).
Usage: trace_concurrency.py <dir with *_kernel_trace.csv>
Prints the time-weighted histogram of the number of kernels executing concurrently, the idle share (no kernel on the
device) and the dispatch rate -- what tells a launch-bound schedule (idle gaps, few kernels at once) from a saturated one."""
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")))
rows.sort()
t_lo = rows[len(rows) // 2][0]
rows = [r for r in rows if r[0] >= t_lo]
ev = []
for s, e, _ in rows:
    ev.append((s, 1))
    ev.append((e, -1))
ev.sort()
hist = {}
cur, last = 0, ev[0][0]
for t, d in ev:
    hist[cur] = hist.get(cur, 0) + (t - last)
    cur += d
    last = t
total = sum(hist.values())
print(f"{len(rows)} kernels in {total / 1e6:.2f} ms: {len(rows) / (total / 1e9) / 1e3:.1f} K dispatches/s")
for k in sorted(hist):
    print(f"  {k:3d} kernels at once: {100.0 * hist[k] / total:5.1f} % of the time")
by = {}
for s, e, n in rows:
    by.setdefault(n, [0, 0])
    by[n][0] += 1
    by[n][1] += e - s
for n, (c, d) in sorted(by.items(), key=lambda kv: -kv[1][1])[:10]:
    print(f"  {n:28s} {c:6d} launches, {d / c / 1e3:9.1f} us average, {d / total:6.2f} x the wall time in sum")
