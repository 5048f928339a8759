#!/bin/bash
# single stream per frame x frames in flight, rotating-owner shard 3/8 of the four stand-ins (GPU_MAX_HW_QUEUES=24 unless noted)
cd $GRAFT_REPO_ROOT
export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29655 HSA_ENABLE_IPC_MODE_LEGACY=0
shard() { python3 bench.py --scene $S --emulate-shard 3/8 --no-cpu-baseline --steps 40 --warmup 8 --force-gather --dist-backend nccl "$@" 2>/dev/null | tail -1 | python3 tools/experiments/print_step.py; }
for S in chess_like street_like atrium_like temple_like; do
  echo "== $S two streams, 8 in flight (shipped)"; shard
  for f in 12 16 18 20; do echo "$S single stream, $f in flight"; PTX_SINGLE_STREAM=1 shard --in-flight $f; done
  echo "$S single stream, 20 in flight, 32 queues"; PTX_SINGLE_STREAM=1 GPU_MAX_HW_QUEUES=32 shard --in-flight 20
done
