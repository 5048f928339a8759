#!/bin/bash
# More reinsertion passes for the tree that is kept, now that a pass costs 5 ms instead of 15 (level lists)?
cd $GRAFT_REPO_ROOT
for n in 32 64 128 256; do
  for s in chess_like atrium_like street_like; do
    PTX_REINSERT=$n python3 bench.py --no-extra-scenes --no-cpu-baseline --scene $s --steps 20 --warmup 3 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('PTX_REINSERT=$n', '$s', 'value %.1f' % d['value'], 'ms/step %.3f' % d['ms_per_step'], 'build %.0f ms' % d['config']['tree_build_ms'], 'closest launch %.4f' % d['roofline']['avg_launch_ms'])"
  done
done
