#!/usr/bin/env python3
"""Turn rocprofv3 PMC passes into per-kernel HBM traffic per launch.

Usage: pmc_traffic.py <fetch_dir> <write_dir> <out.json> [scene [shape]]   (scene: bench.py --scene of the profiled command, default
       chess_like; shape: bench.py's shape key of it, WxH/SPPspp/dDEPTH/shardRofN, default 1920x1080/8spp/d8/shard0of1)
where the dirs hold the counter_collection.csv of two separate passes of the SAME command,
  rocprofv3 --kernel-trace --pmc FETCH_SIZE  --output-format csv -d <fetch_dir> -- python3 bench.py ...
  rocprofv3 --kernel-trace --pmc WRITE_SIZE  --output-format csv -d <write_dir> -- python3 bench.py ...
(FETCH_SIZE takes 3 TCC slots and WRITE_SIZE 2 of the 4, so they cannot share a pass.)

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md "HBM": both counters are in KiB, and on gfx950 FETCH_SIZE
reports exactly half of the bytes of a wide COALESCED streaming read (128-byte requests tallied at 64 B).  What it reports for this
repository's other pattern -- every lane fetching its OWN 64-byte node or 48-byte triangle -- is measured by
tools/experiments/fetch_size_calibration.hip (profiles/r05_fetch_size_calibration.txt, round 5): counter / known bytes = 0.500 for the
coalesced float4 stream, 1.000 for one 64-byte record per lane (four dwordx4, every record once; 1.000 of the REQUESTED bytes when
every record is read twice), 1.667 for 48-byte records at a 48-byte stride (= the 64-byte lines they touch: 1.667 lines per record).
So the factor is per kernel (FETCH_FACTOR below):
  2   streaming kernels (path-state streams, 16 B per lane, coalesced): k_generate, k_accumulate, k_apply_shadow, k_copy_out, fills, copies
  1   traversal kernels (divergent node / triangle records; their coalesced share -- queue entries, rays -- is under-counted by
      at most the 56 B per ray of DESIGN.md section 5): k_trace_closest, k_trace_shadow, k_tail, k_trace_rays, k_sample_tree_cost
  1   k_shade, as a LOWER bound: divergent 272-byte shading records and texels (factor 1) beside coalesced path state (factor 2);
      `hbm_bytes_per_launch_upper` carries the factor-2 figure of every kernel (what rounds 1-4 reported)
hbm_bytes = (factor * FETCH_SIZE + WRITE_SIZE) * 1024.
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


STREAMING = ("k_generate", "k_accumulate", "k_apply_shadow", "k_copy_out", "__amd_rocclr", "k_restart", "k_finish_restarts", "k_prologue",
             "k_upload_lights", "k_pack_shard", "k_unpack_shard", "k_snapshot")


def fetch_factor(kernel):
    """Measured (profiles/r05_fetch_size_calibration.txt): 2 for coalesced 16-B-per-lane streams, 1 for per-lane records."""
    return 2.0 if kernel.startswith(STREAMING) else 1.0


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
        kk = k.replace("ptd::", "")
        out[k] = {"launches": n, "fetch_size_kib_per_launch": f_kib, "write_size_kib_per_launch": w_kib, "fetch_factor": fetch_factor(kk),
                  "hbm_bytes_per_launch": (fetch_factor(kk) * f_kib + w_kib) * 1024.0, "hbm_bytes_per_launch_upper": (2.0 * f_kib + w_kib) * 1024.0}
    ranked = sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])
    out["calibration"] = "profiles/r05_fetch_size_calibration.txt: FETCH_SIZE / known bytes = 0.500 coalesced float4 stream, 1.000 one 64-B record per lane"
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
