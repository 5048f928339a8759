#!/bin/bash
# tools/profile_set.sh TAG [--scene NAME] [bench.py args ...] -- run ON THE GPU BOX (through gpurun): the profile set of one build.
#   gpurun_out/TAG_kernel_stats.csv   rocprofv3 --kernel-trace --stats of the default bench.py workload
#   gpurun_out/TAG_kernel_stats_one_in_flight.csv   the same with --in-flight 1
#   gpurun_out/TAG_traffic.json       per-kernel HBM bytes per launch from two PMC passes (FETCH_SIZE, WRITE_SIZE; separate
#                                     passes, no trace domains beside --kernel-trace, as the guide prescribes)
#   gpurun_out/TAG_sq.txt / .json     SQ counter summary (wave cycles / waiting / issuing, instruction mix)
#   gpurun_out/TAG_bench.json         the un-profiled bench.py line of the same build
# Copy what is to be judged into profiles/ afterwards.
TAG=${1:?tag}; shift
SCENE=chess_like; W=1920; H=1080; SPP=8; DEPTH=8; SHARD=0/1; prev=""
for a in "$@"; do
  [ "$prev" = "--scene" ] && SCENE=$a; [ "$prev" = "--width" ] && W=$a; [ "$prev" = "--height" ] && H=$a; [ "$prev" = "--spp" ] && SPP=$a
  [ "$prev" = "--depth" ] && DEPTH=$a; [ "$prev" = "--shard" ] && SHARD=$a; prev=$a
done
SHAPE="${W}x${H}/${SPP}spp/d${DEPTH}/shard${SHARD%/*}of${SHARD#*/}"   # bench.py shape_key()
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
mkdir -p gpurun_out
CMD="python3 bench.py --steps 5 --warmup 1 --repeats 1 --no-cpu-baseline --no-extra-scenes $*"
# the kernel trace runs the default step count, so that its per-kernel average is taken over the same mix of launches as the
# live average of the un-profiled line (few steps weigh the host-driven learning frames of each renderer too much)
TRACE_CMD="python3 bench.py --repeats 1 --no-cpu-baseline --no-extra-scenes $*"
for d in trace fetch write sq1 sq2; do rm -rf gpurun_out/${TAG}_$d; done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_trace -o $TAG -- $TRACE_CMD > gpurun_out/${TAG}_trace.log 2>&1; echo "trace rc=$?"
# the same workload with ONE frame in flight: what bench.py's top-level roofline measures live (launch durations without other
# frames' kernels on the machine; tracing perturbs the overlap of the default command, not this)
rm -rf gpurun_out/${TAG}_trace1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_trace1 -o $TAG -- $TRACE_CMD --in-flight 1 > gpurun_out/${TAG}_trace1.log 2>&1; echo "trace1 rc=$?"
cp "$(find gpurun_out/${TAG}_trace1 -name '*kernel_stats.csv' | head -1)" gpurun_out/${TAG}_kernel_stats_one_in_flight.csv 2>/dev/null
find gpurun_out/${TAG}_trace1 -name "*.csv" -size +1M -delete
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/${TAG}_fetch -o $TAG -- $CMD > gpurun_out/${TAG}_fetch.log 2>&1; echo "fetch rc=$?"
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/${TAG}_write -o $TAG -- $CMD > gpurun_out/${TAG}_write.log 2>&1; echo "write rc=$?"
python3 tools/pmc_traffic.py gpurun_out/${TAG}_fetch gpurun_out/${TAG}_write gpurun_out/${TAG}_traffic.json $SCENE $SHAPE
python3 tools/trace_gaps.py gpurun_out/${TAG}_trace | grep step | tail -3
CMD3="python3 bench.py --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-extra-scenes $*"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/${TAG}_sq1 -o sq -- $CMD3 > gpurun_out/${TAG}_sq1.log 2>&1; echo "sq1 rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d gpurun_out/${TAG}_sq2 -o sq -- $CMD3 > gpurun_out/${TAG}_sq2.log 2>&1; echo "sq2 rc=$?"
python3 tools/pmc_sq.py --json gpurun_out/${TAG}_sq.json --scene $SCENE --shape $SHAPE --stats-alone gpurun_out/${TAG}_kernel_stats_one_in_flight.csv gpurun_out/${TAG}_sq1 gpurun_out/${TAG}_sq2 > gpurun_out/${TAG}_sq.txt 2>&1
cp "$(find gpurun_out/${TAG}_trace -name '*kernel_stats.csv' | head -1)" gpurun_out/${TAG}_kernel_stats.csv 2>/dev/null
find gpurun_out/${TAG}_trace gpurun_out/${TAG}_fetch gpurun_out/${TAG}_write gpurun_out/${TAG}_sq1 gpurun_out/${TAG}_sq2 -name "*.csv" -size +1M -delete
# the un-profiled line quotes the traffic of THIS build: it looks for the newest profiles/r*_traffic.json
cp gpurun_out/${TAG}_traffic.json profiles/${TAG}_traffic.json
cp gpurun_out/${TAG}_sq.json profiles/${TAG}_sq.json
timeout 900 python3 bench.py $* > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; tail -c 2500 gpurun_out/${TAG}_bench.json
