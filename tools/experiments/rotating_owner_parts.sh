#!/bin/bash
# Which part of a rotating-owner step costs what (1 / 8 shard of chess_like, 8 frames in flight, one GPU)
cd $GRAFT_REPO_ROOT
export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29655 HSA_ENABLE_IPC_MODE_LEGACY=0
S=${SCENE:-chess_like}
run() { python3 bench.py --scene $S --emulate-shard ${SHARD:-0/8} --no-cpu-baseline --steps 40 --warmup 8 "$@" 2>/dev/null | tail -1 | python3 tools/experiments/print_step.py; }
G="--force-gather --dist-backend nccl"
echo "== $S"
echo "shard alone";                                             run
echo "rank 1 of a rank0-owned job WITHOUT the collective call: render into the bound send buffer only"; SHARD=1/8 run $G --root rank0 --emulate-collective off
echo "rank 1 of a rank0-owned job: render + collective call, never the owner";   SHARD=1/8 run $G --root rank0
echo "the same with all_gather";                                 SHARD=1/8 run $G --root rank0 --collective all_gather
echo "rotating owner, no read-back (owner's steps: device image)"; run $G --emulate-readback off
echo "rotating owner";                                           run $G
echo "rotating owner, 12 frames in flight";                      run $G --in-flight 12
echo "rotating owner, 16 frames in flight";                      run $G --in-flight 16
echo "rotating owner, 6 frames in flight";                       run $G --in-flight 6
