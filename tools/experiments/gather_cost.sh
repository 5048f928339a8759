#!/bin/bash
# What rank 0's gather duty costs a 1/8 shard step (chess_like, 8 frames in flight), piece by piece.
cd $GRAFT_REPO_ROOT
export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29655 HSA_ENABLE_IPC_MODE_LEGACY=0
run() { python3 bench.py --scene ${SCENE:-chess_like} --emulate-shard 0/8 --no-cpu-baseline --steps 40 --warmup 8 "$@" 2>/dev/null | tail -1 | python3 tools/experiments/print_step.py; }
echo "shard alone";                          run
echo "shard + read-back";                    run --emulate-readback on
echo "shard + gather duty, no read-back";    run --force-gather --dist-backend nccl --emulate-readback off
echo "shard + gather duty + read-back (ptx_unpack_shard + ptx_readback_begin, rounds 1-4)"; run --force-gather --dist-backend nccl --gather-readback separate
echo "shard + gather duty + read-back (ptx_unpack_shard_host: the frame goes to the host inside the unpack kernels)"; run --force-gather --dist-backend nccl --gather-readback fused
echo "whole frame (N = 1), read-back, PTX_COPY_GROUPS=1 / 4"; 
for g in 1 4; do PTX_COPY_GROUPS=$g python3 bench.py --scene chess_like --emulate-shard 0/1 --emulate-readback on --no-cpu-baseline --steps 20 --warmup 4 2>/dev/null | tail -1 | python3 tools/experiments/print_step.py; done
