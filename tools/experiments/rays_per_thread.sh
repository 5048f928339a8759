#!/bin/bash
# Rays per persistent traversal thread when ONE frame is in flight (the launch-alone leg of bench.py; a host with one renderer)
cd $GRAFT_REPO_ROOT
for n in default 1 2 4 8; do
  for s in chess_like atrium_like; do
    if [ $n = default ]; then unset PTX_RAYS_PER_THREAD; else export PTX_RAYS_PER_THREAD=$n; fi
    python3 bench.py --no-extra-scenes --no-cpu-baseline --scene $s --steps 10 --warmup 2 --in-flight 1 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); rf=d['roofline']; print('rays/thread $n', '$s', 'in-flight 1: value %.1f' % d['value'], 'ms/step %.3f' % d['ms_per_step'], 'closest launch alone %.4f' % rf['avg_launch_ms'], 'frac %.3f' % rf['frac'])"
  done
done
