#!/usr/bin/env python3
"""Build libptx_hip.so with -Rpass-analysis=kernel-resource-usage and print one line per kernel: registers, spills,
scratch, LDS, occupancy.  (The same flags as path_tracing_amd.build(); the library written is the product library.)
Usage: tools/kernel_resources.py [filter-substring] [-o other.so] [-- extra hipcc flags]
(-o: an experimental build beside the product library, to be loaded through PTX_HIP_LIB)"""
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import __graft_entry__ as graft  # noqa: E402


def main():
    args = sys.argv[1:]
    extra = []
    if "--" in args:
        k = args.index("--")
        args, extra = args[:k], args[k + 1:]
    out = None
    if "-o" in args:
        k = args.index("-o")
        out = args[k + 1]
        args = args[:k] + args[k + 2:]
    flt = args[0] if args else ""
    pkg = graft.load_package()
    src = pkg.hip_sources()[0]
    cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + pkg.HIPCC_FLAGS + extra + ["-Rpass-analysis=kernel-resource-usage", "-o", out or pkg.HIP_LIB, src]
    p = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
    cur, rows = None, []
    for line in p.stderr.splitlines():
        m = re.search(r"remark: (?:Function )?Name: (\S+)", line)
        if m:
            cur = {"name": subprocess.run(["c++filt", m.group(1)], stdout=subprocess.PIPE, text=True).stdout.strip().split("(")[0]}
            rows.append(cur)
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+(?:\[[^\]]*\])?):\s+(\d+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
        elif "error" in line:
            print(line)
    print(f"{'kernel':44s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'vspill':>6s} {'sspill':>6s} {'scratch':>7s} {'LDS':>6s} {'waves':>5s}")
    for r in rows:
        if flt in r["name"]:
            print(f"{r['name'][:44]:44s} {r.get('VGPRs', 0):5d} {r.get('AGPRs', 0):5d} {r.get('SGPRs', 0):5d} {r.get('VGPRs Spill', 0):6d} "
                  f"{r.get('SGPRs Spill', 0):6d} {r.get('ScratchSize [bytes/lane]', 0):7d} {r.get('LDS Size [bytes/block]', 0):6d} "
                  f"{r.get('Occupancy [waves/SIMD]', 0):5d}")
    sys.exit(p.returncode)


if __name__ == "__main__":
    main()
