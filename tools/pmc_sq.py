#!/usr/bin/env python3
"""Per-kernel sums of SQ counters from rocprofv3 --pmc passes (counter_collection.csv), per launch.
Usage: pmc_sq.py <dir> [<dir> ...]   (each dir = one pass; counters of all passes are merged by kernel name)"""
import collections
import csv
import glob
import sys


def main():
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for d in sys.argv[1:]:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].split("(")[0]
                c = agg[k][r["Counter_Name"]]
                c[0] += 1
                c[1] += float(r["Counter_Value"])
    for k, counters in sorted(agg.items()):
        if not k.startswith("k_") and not k.startswith("void k_") and "k_" not in k:
            continue
        print(k)
        for name, (n, v) in sorted(counters.items()):
            print(f"    {name:28s} {v / n:16.1f} per launch ({n} launches)")


if __name__ == "__main__":
    main()
