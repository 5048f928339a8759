#!/bin/bash
# PTX_SINGLE_STREAM (round 6): one stream per frame in flight instead of two -- twice the frames on the same hardware queues
cd $GRAFT_REPO_ROOT
export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29655 HSA_ENABLE_IPC_MODE_LEGACY=0
shard() { python3 bench.py --scene chess_like --emulate-shard 3/8 --no-cpu-baseline --steps 40 --warmup 8 --force-gather --dist-backend nccl "$@" 2>/dev/null | tail -1 | python3 tools/experiments/print_step.py; }
frame() { python3 bench.py --scene $1 --no-extra-scenes --no-cpu-baseline --steps 20 --warmup 4 --min-seconds 1.5 --in-flight $2 > /dev/null 2>&1; python3 -c "import json; d=json.load(open('bench_detail.json')); print('   value %.1f  ms/step %.3f' % (d['value'], d['ms_per_step']))"; }
echo "two streams per frame, 8 in flight (shipped): rotating-owner shard 3/8"; shard
for f in 8 12 16 24; do for q in 24 32; do echo "single stream, $f in flight, $q queues: shard 3/8"; PTX_SINGLE_STREAM=1 GPU_MAX_HW_QUEUES=$q shard --in-flight $f; done; done
for s in chess_like atrium_like; do
  echo "$s whole frame, two streams, 8 in flight"; frame $s 8
  for f in 8 12 16; do echo "$s whole frame, single stream, $f in flight"; PTX_SINGLE_STREAM=1 frame $s $f; done
done
