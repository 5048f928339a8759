#!/usr/bin/env python3
"""GPU idle time inside the timed steps of a rocprofv3 --kernel-trace run.
Usage: trace_gaps.py <dir with *_kernel_trace.csv>
Merges the busy intervals of all kernels (all streams), finds the steps as the spans between consecutive
k_generate launches, and reports per step: wall span, union busy time, idle time, and the largest gaps with
the kernels on either side."""
import csv
import glob
import sys


def main():
    rows = []
    for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]))
    rows.sort()
    gens = [i for i, r in enumerate(rows) if "k_generate" in r[2] and "mip" not in r[2]]
    for a, b in zip(gens[1:-1], gens[2:]):  # skip the warm-up step
        step = rows[a:b]
        t0, t1 = step[0][0], max(r[1] for r in step)
        busy, cur_s, cur_e = 0, step[0][0], step[0][1]
        gaps = []
        last_name = step[0][2]
        for s, e, name in step[1:]:
            if s > cur_e:
                busy += cur_e - cur_s
                gaps.append((s - cur_e, last_name, name))
                cur_s, cur_e = s, e
            else:
                cur_e = max(cur_e, e)
            last_name = name
        busy += cur_e - cur_s
        gaps.sort(reverse=True)
        print(f"step: span {(t1 - t0) / 1e6:.3f} ms, busy {busy / 1e6:.3f} ms, idle {(t1 - t0 - busy) / 1e6:.3f} ms in {len(gaps)} gaps; kernels {len(step)}")
        for g, x, y in gaps[:6]:
            print(f"    {g / 1e3:8.1f} us between {x} -> {y}")


if __name__ == "__main__":
    main()
