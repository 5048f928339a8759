#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { # label env... -- args
  python3 bench.py --emulate-shard 0/8 --no-cpu-baseline --steps 40 --warmup 8 "$@" 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('  ms/step %.4f' % d['ms_per_step'], d['config']['kernel_ms_per_step'])"
}
for scene in chess_like street_like; do
for t in 20000 37500 75000 150000 300000 600000; do
  echo "$scene tail threshold $t in-flight 8"; PTX_TAIL_THRESHOLD=$t run --scene $scene
done
for f in 4 6 12 16; do
  echo "$scene tail threshold default in-flight $f"; run --scene $scene --in-flight $f
done
done
