#!/usr/bin/env python3
"""Print per-dispatch PMC counters of one kernel from rocprofv3 counter_collection.csv files."""
import collections, csv, glob, sys
kernel = sys.argv[1]
rows = collections.defaultdict(dict)
for d in sys.argv[2:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        per = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel_Name"]:
                per[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
        for c, v in per.items():
            for i, (_, val) in enumerate(sorted(v)):
                rows[i][c] = val
names = sorted({c for r in rows.values() for c in r})
for i in sorted(rows):
    print(f"dispatch {i}: " + "  ".join(f"{c}={rows[i].get(c, float('nan')):.4g}" for c in names))
