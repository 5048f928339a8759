#!/bin/bash
# More hardware queues than 16, more frames in flight than 8?  (GPU_MAX_HW_QUEUES is read by the HIP runtime at start-up.)
cd $GRAFT_REPO_ROOT
for q in 16 24 32; do for f in 8 12 16; do
  echo "GPU_MAX_HW_QUEUES=$q in-flight $f"
  for s in chess_like atrium_like; do
    GPU_MAX_HW_QUEUES=$q python3 bench.py --no-extra-scenes --no-cpu-baseline --scene $s --steps 24 --warmup 4 --in-flight $f 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('   ', d['config']['workload'].split()[0], 'value %.1f' % d['value'], 'ms/step %.3f' % d['ms_per_step'])"
  done
done; done
