#!/usr/bin/env python3
"""FETCH_SIZE (KiB) per launch of tools/experiments/fetch_size_calibration.hip against the bytes each kernel is known to read, and the
rate each pattern reaches (kernel durations of the same run's kernel trace): the streaming rate and the rate of independent
per-lane 64-byte gathers -- the ceiling a traversal could reach if its fetches were not a dependent chain.
Usage: fetch_size_calibration.py <rocprofv3 output dir>"""
import collections
import csv
import glob
import sys

KNOWN = {"k_stream": 1 << 30, "k_gather64": None, "k_gather48": (1 << 24) * 48}
rows = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE":
            rows[r["Kernel_Name"].split("(")[0]].append((int(r.get("Grid_Size", 0) or 0), float(r["Counter_Value"])))
for k, v in sorted(rows.items()):
    for grid, kib in sorted(set(v)):
        known = KNOWN.get(k)
        what = k
        if k == "k_gather64":
            twice = grid > (1 << 24)
            known = (1 << 24) * 64 * (2 if twice else 1)
            what = "k_gather64 (every record twice: requested bytes)" if twice else "k_gather64"
        if known:
            print(f"{what:52s} FETCH_SIZE {kib / 1024:9.1f} MiB   known {known / 2**20:8.1f} MiB   counter / known = {kib * 1024 / known:.3f}")

dur = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        dur[(k, int(r.get("Grid_Size_X", 0) or 0))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9)
for (k, grid), v in sorted(dur.items()):
    if k not in KNOWN:
        continue
    known = KNOWN[k] if KNOWN[k] else (1 << 24) * 64 * (2 if grid > (1 << 24) else 1)
    best = min(v)
    print(f"{k:12s} grid {grid:9d}: {known / 2**30:5.2f} GiB in {best * 1e6:7.1f} us (best of {len(v)}) = {known / best / 1e12:5.2f} TB/s")
