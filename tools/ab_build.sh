#!/bin/bash
# tools/ab_build.sh NAME [extra hipcc flags...] -- an experimental build of the HIP library (same ABI) as tools/ab/libptx_NAME.so;
# run it with PTX_HIP_LIB=tools/ab/libptx_NAME.so (tools/ab_run.sh).  tools/ab/ travels to the GPU box; *.so stays out of git.
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
mkdir -p tools/ab
FLAGS=$(python3 -c "
import __graft_entry__ as g
print(' '.join(g.load_package().HIPCC_FLAGS))")
/opt/rocm/bin/hipcc $FLAGS "$@" -o tools/ab/libptx_$NAME.so path-tracing_amd/csrc/ptx_capi.hip
ls -la tools/ab/libptx_$NAME.so
