#!/usr/bin/env python3
"""Print the kernel timeline of the LAST step of a rocprofv3 --kernel-trace run (start offset, duration, name).
Usage: trace_timeline.py <dir>"""
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", ""), r.get("Stream_Id", r.get("Queue_Id", "?"))))
rows.sort()
gens = [i for i, r in enumerate(rows) if r[2].startswith("k_generate") and "mip" not in r[2]]
a, b = gens[-2], gens[-1]
t0 = rows[a][0]
for s, e, name, q in rows[a:b]:
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  q{q}  {name}")
