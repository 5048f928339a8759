#!/bin/bash
# Does the mere presence of an RCCL communicator in the process cost a 1 / 8 shard step?  (render into the bound send buffer, no
# collective call, never the owner: the only difference between the lines is the process group)
cd $GRAFT_REPO_ROOT
export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29655 HSA_ENABLE_IPC_MODE_LEGACY=0
S=${SCENE:-chess_like}
run() { python3 bench.py --scene $S --emulate-shard 1/8 --no-cpu-baseline --steps 40 --warmup 8 --root rank0 --emulate-collective off "$@" 2>/dev/null | tail -1 | python3 tools/experiments/print_step.py; }
echo "no process group (shard alone, row-major accumulation)";  run
echo "gloo group, bound accumulation";                 run --force-gather --dist-backend gloo
echo "nccl group, bound accumulation";                 run --force-gather --dist-backend nccl
echo "nccl group, packed accumulation (pack launch)";  run --force-gather --dist-backend nccl --shard-accumulation packed
for q in 20 24 32; do echo "nccl group, GPU_MAX_HW_QUEUES=$q"; GPU_MAX_HW_QUEUES=$q run --force-gather --dist-backend nccl; done
echo "nccl group, 7 frames in flight";                 run --force-gather --dist-backend nccl --in-flight 7
echo "nccl group, 16 frames in flight";                run --force-gather --dist-backend nccl --in-flight 16
echo "no group, 16 frames in flight";                  run --in-flight 16
