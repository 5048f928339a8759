#!/bin/bash
# Kernel trace of a 1 / 8 shard (rendering alone, eight frames in flight and one): per-kernel averages, concurrency, gaps
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for f in 8 1; do
  rm -rf gpurun_out/shard_trace_$f
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/shard_trace_$f -o st -- python3 bench.py --scene chess_like --emulate-shard 3/8 --no-cpu-baseline --steps 40 --warmup 8 --repeats 1 --in-flight $f > gpurun_out/shard_trace_$f.log 2>&1
  echo "== in flight $f"; tail -1 gpurun_out/shard_trace_$f.log | python3 tools/experiments/print_step.py
  python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/shard_trace_$f/**/*kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:14]: print("   %-42s calls %6s avg %8.1f us  %5.1f %%" % (r["Name"][:42], r["Calls"], float(r["AverageNs"])/1e3, 100*float(r["TotalDurationNs"])/tot))
PY
  python3 tools/trace_concurrency.py gpurun_out/shard_trace_$f | head -14
  python3 tools/trace_gaps.py gpurun_out/shard_trace_$f | grep step | tail -2
  find gpurun_out/shard_trace_$f -name "*.csv" -size +1M -delete
done
