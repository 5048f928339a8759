#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
for sh in 0/8 0/1; do
  d=gpurun_out/tl_${sh/\//_}
  rm -rf $d
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $d -- python3 bench.py --steps 4 --warmup 2 --emulate-shard $sh > $d.log 2>&1
  echo "== shard $sh"; tail -1 $d.log | cut -c1-300
  python3 tools/trace_timeline.py $d > $d.timeline.txt
  python3 tools/trace_gaps.py $d | tail -25
done
