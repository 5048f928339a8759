#!/usr/bin/env python3
"""Per-kernel sums of SQ counters from rocprofv3 --pmc passes (counter_collection.csv), per launch.
Usage: pmc_sq.py [--json out.json --scene NAME --shape KEY --stats-alone kernel_stats_one_in_flight.csv] <dir> [<dir> ...]
Each dir = one pass; counters of all passes are merged by kernel name (template variants kept apart).  The text summary goes to
stdout; --json also writes {kernel: {counter: value per launch, "launches": n}, "scene", "shape", "source_digest"}, which
bench.py reads for roofline.shade (VALU wave-instructions per launch of k_shade against the chip's issue rate).  --stats-alone: the
rocprofv3 --kernel-trace --stats summary of the same command with ONE frame in flight; every kernel's average duration there goes
into the JSON as "rocprof_ms_alone" (the kernel's own duration: bench.py's HIP-event bracket of k_shade also holds the wait for the
previous bounce's k_apply_shadow on the other stream)."""
import collections
import csv
import glob
import hashlib
import json
import os
import sys


def source_digest():
    csrc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "path-tracing_amd", "csrc")
    h = hashlib.sha256()
    for f in [os.path.join(csrc, "ptx_capi.hip")] + sorted(glob.glob(os.path.join(csrc, "*.hpp"))):  # = path-tracing_amd.hip_sources()
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def main():
    args = sys.argv[1:]
    opts = {"--json": None, "--scene": "chess_like", "--shape": "1920x1080/8spp/d8/shard0of1", "--stats-alone": None}
    while args and args[0] in opts:
        opts[args[0]] = args[1]
        args = args[2:]
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for d in args:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].split("(")[0]
                c = agg[k][r["Counter_Name"]]
                c[0] += 1
                c[1] += float(r["Counter_Value"])
    doc = {}
    for k, counters in sorted(agg.items()):
        if not k.startswith("k_") and not k.startswith("void k_") and "k_" not in k:
            continue
        print(k)
        entry = {}
        for name, (n, v) in sorted(counters.items()):
            print(f"    {name:28s} {v / n:16.1f} per launch ({n} launches)")
            entry[name] = v / n
            entry["launches"] = n
        doc[k.replace("void ", "")] = entry
    if opts["--stats-alone"] and os.path.exists(opts["--stats-alone"]):
        for r in csv.DictReader(open(opts["--stats-alone"])):
            k = r["Name"].split("(")[0].replace("void ", "")
            if k in doc:
                doc[k]["rocprof_ms_alone"] = float(r["AverageNs"]) * 1e-6
                doc[k]["rocprof_calls_alone"] = int(r["Calls"])
    if opts["--json"]:
        doc["scene"], doc["shape"], doc["source_digest"] = opts["--scene"], opts["--shape"], source_digest()
        json.dump(doc, open(opts["--json"], "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
