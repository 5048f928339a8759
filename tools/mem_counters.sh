#!/bin/bash
# tools/mem_counters.sh [SCENE] -- run ON THE GPU BOX: memory-pipeline counters (per-CU TLB, texture addresser, L1, L2 hits / misses) of the render
# kernels with one frame in flight, six separate --pmc passes (no trace domains beside --kernel-trace) -> gpurun_out/mem_SCENE.txt
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
SC=${1:-chess_like}
CMD="python3 bench.py --scene $SC --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-extra-scenes --in-flight 1"
i=0
for set in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" \
           "TA_TA_BUSY_sum GRBM_GUI_ACTIVE" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1)); rm -rf gpurun_out/mem_${SC}_$i
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/mem_${SC}_$i -o m -- $CMD > gpurun_out/mem_${SC}_$i.log 2>&1; echo "pass $i rc=$?"
  if grep -q "exceeds the capabilities" gpurun_out/mem_${SC}_$i.log; then echo "pass $i: too many counters"; fi
done
python3 tools/pmc_sq.py gpurun_out/mem_${SC}_1 gpurun_out/mem_${SC}_2 gpurun_out/mem_${SC}_3 gpurun_out/mem_${SC}_4 gpurun_out/mem_${SC}_5 gpurun_out/mem_${SC}_6 > gpurun_out/mem_${SC}.txt 2>&1
find gpurun_out/ -path "*mem_${SC}_*" -name "*.csv" -size +1M -delete
