#!/usr/bin/env python3
"""Turn rocprofv3 PMC passes into per-kernel HBM traffic per launch.

Usage: pmc_traffic.py <fetch_dir> <write_dir> <out.json> [scene [shape]]   (scene: bench.py --scene of the profiled command, default
       chess_like; shape: bench.py's shape key of it, WxH/SPPspp/dDEPTH/shardRofN, default 1920x1080/8spp/d8/shard0of1)
where the dirs hold the counter_collection.csv of two separate passes of the SAME command,
  rocprofv3 --kernel-trace --pmc FETCH_SIZE  --output-format csv -d <fetch_dir> -- python3 bench.py ...
  rocprofv3 --kernel-trace --pmc WRITE_SIZE  --output-format csv -d <write_dir> -- python3 bench.py ...
(FETCH_SIZE takes 3 TCC slots and WRITE_SIZE 2 of the 4, so they cannot share a pass.)

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md "HBM": both counters are
in KiB; on gfx950 FETCH_SIZE reports exactly half of the bytes of 16-B-per-lane reads, so
hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.  The factor 2 is calibrated by the guide for
coalesced dwordx4 streams; our traversal reads are dwordx4 per lane but scattered, so the
absolute value is an upper estimate (the guide: other patterns are uncalibrated).
"""
import collections
import csv
import glob
import hashlib
import json
import os
import sys


def source_digest():
    """Same digest as bench.py's: which kernels these counters were collected on."""
    csrc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "path-tracing_amd", "csrc")
    h = hashlib.sha256()
    for f in [os.path.join(csrc, "ptx_capi.hip")] + sorted(glob.glob(os.path.join(csrc, "*.hpp"))):  # = path-tracing_amd.hip_sources()
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def load(d):
    files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in files:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]  # template variants of one kernel merge
            agg[k][0] += 1
            agg[k][1] += float(r["Counter_Value"])
    return agg


def main():
    fetch, write = load(sys.argv[1]), load(sys.argv[2])
    out = {}
    for k in sorted(set(fetch) | set(write)):
        nf, vf = fetch.get(k, [0, 0.0])
        nw, vw = write.get(k, [0, 0.0])
        n = max(nf, nw)
        if not n:
            continue
        f_kib, w_kib = (vf / nf if nf else 0.0), (vw / nw if nw else 0.0)
        out[k] = {"launches": n, "fetch_size_kib_per_launch": f_kib, "write_size_kib_per_launch": w_kib,
                  "hbm_bytes_per_launch": (2.0 * f_kib + w_kib) * 1024.0}
    ranked = sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])
    out["source_digest"] = source_digest()
    out["scene"] = sys.argv[4] if len(sys.argv) > 4 else "chess_like"
    out["shape"] = sys.argv[5] if len(sys.argv) > 5 else "1920x1080/8spp/d8/shard0of1"
    json.dump(out, open(sys.argv[3], "w"), indent=1, sort_keys=True)
    for k, v in ranked[:8]:
        if not isinstance(v, dict):
            continue
        print(f"{k:28s} launches {v['launches']:4d}  {v['hbm_bytes_per_launch'] / 1e6:10.2f} MB/launch")


if __name__ == "__main__":
    main()
