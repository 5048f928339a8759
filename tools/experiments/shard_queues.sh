#!/bin/bash
# frames in flight x hardware queues for a rotating-owner 1 / 8 shard step (a gathering job runs 2 streams per frame + torch's + RCCL's)
cd $GRAFT_REPO_ROOT
export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29655 HSA_ENABLE_IPC_MODE_LEGACY=0
S=${SCENE:-chess_like}
run() { python3 bench.py --scene $S --emulate-shard ${SHARD:-3/8} --no-cpu-baseline --steps 40 --warmup 8 --force-gather --dist-backend nccl "$@" 2>/dev/null | tail -1 | python3 tools/experiments/print_step.py; }
for f in 8 10 12 16; do for q in 24 32 40; do echo "in flight $f, GPU_MAX_HW_QUEUES=$q"; GPU_MAX_HW_QUEUES=$q run --in-flight $f; done; done
