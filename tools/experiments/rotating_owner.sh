#!/bin/bash
# Round 6: what one rank's share of an 8-GPU step costs (chess_like, 8 frames in flight) as the gather duty moves from "rank 0,
# every step, 8 unpack launches, device image + host frame" to "owner k % 8, one launch, host-only stores, accumulation in the
# gather's message".  One GPU; the pieces that would arrive over xGMI are one device copy on the owner's steps.
cd $GRAFT_REPO_ROOT
export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29655 HSA_ENABLE_IPC_MODE_LEGACY=0
S=${SCENE:-chess_like}
run() { python3 bench.py --scene $S --emulate-shard ${SHARD:-0/8} --no-cpu-baseline --steps 40 --warmup 8 "$@" 2>/dev/null | tail -1 | python3 tools/experiments/print_step.py; }
G="--force-gather --dist-backend nccl"
echo "== $S"
echo "whole frame (N = 1) with read-back";   SHARD=0/1 run --emulate-readback on --steps 20 --warmup 4
echo "shard alone";                          run
echo "round 5: rank 0 owns every frame, 8 ptx_unpack_shard_host launches, pack"; run $G --root rank0 --gather-unpack per-rank --shard-accumulation packed
echo "  + one ptx_unpack_shards launch (host-only stores)";                      run $G --root rank0 --shard-accumulation packed
echo "  + accumulation bound to the send buffer (no pack)";                     run $G --root rank0
echo "  + rotating owner (shard 0: owns steps 0, 8, 16 ...)";                    run $G
for g in 4 8 32 64; do echo "    rotating owner, PTX_COPY_GROUPS=$g"; PTX_COPY_GROUPS=$g run $G; done
echo "rotating owner, every shard:"
for r in 0 1 2 3 4 5 6 7; do SHARD=$r/8 run $G; done
