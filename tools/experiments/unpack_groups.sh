#!/bin/bash
# Workgroups of the unpack-to-host kernels of the fused gather read-back (PTX_COPY_GROUPS), 1 / 8 shard of chess_like, 8 frames in flight
cd $GRAFT_REPO_ROOT
export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29655 HSA_ENABLE_IPC_MODE_LEGACY=0
for g in 2 4 8 16 32 64 2048; do
  echo "PTX_COPY_GROUPS=$g"
  PTX_COPY_GROUPS=$g python3 bench.py --scene ${SCENE:-chess_like} --emulate-shard 0/8 --no-cpu-baseline --steps 40 --warmup 8 --force-gather --dist-backend nccl 2>/dev/null | tail -1 | python3 tools/experiments/print_step.py
done
