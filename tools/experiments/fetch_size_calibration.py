#!/usr/bin/env python3
"""FETCH_SIZE (KiB) per launch of tools/experiments/fetch_size_calibration.hip against the bytes each kernel is known to read.
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
