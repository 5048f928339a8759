#!/bin/bash
# Who runs where (round 6): the streams of the frames in flight created with CU masks (PTX_CU_PARTITION, pt_runtime.hpp)
#   0 none   1 one XCD per frame in flight   2 two groups of four XCDs   3 main streams on XCDs 0-5, auxiliary streams on 6-7
#   4 two XCDs per frame   5 main stream on XCD h, auxiliary stream on XCD h + 4
#   controls: 6 masked streams with every CU enabled   7 the library's own plain streams instead of the host's
cd $GRAFT_REPO_ROOT
val() { python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('   value %.1f  ms/step %.3f' % (d['value'], d['ms_per_step']))"; }
for s in ${SCENES:-chess_like atrium_like temple_like street_like}; do
  for m in ${MODES:-0 1 2 3 4 5 6 7}; do
    echo "$s PTX_CU_PARTITION=$m whole frame"
    PTX_CU_PARTITION=$m python3 bench.py --scene $s --no-extra-scenes --no-cpu-baseline --steps 20 --warmup 4 --min-seconds 1.5 2>/dev/null | tail -1 | val
  done
done
for m in ${MODES:-0 1 2 3 4 5 6 7}; do
  echo "chess_like PTX_CU_PARTITION=$m shard 3/8 alone"
  PTX_CU_PARTITION=$m python3 bench.py --scene chess_like --emulate-shard 3/8 --no-cpu-baseline --steps 40 --warmup 8 2>/dev/null | tail -1 | python3 tools/experiments/print_step.py
done
