#!/bin/bash
# tools/profile_all.sh TAG -- the profile sets of one build for the headline, the other stand-ins at its shape and BASELINE's configs at
# their own shapes (tools/profile_set.sh each; run ON THE GPU BOX).  PART=a|b splits the work over two gpurun calls.
TAG=${1:?tag}; PART=${2:-ab}
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
run() { t=$1; shift; tools/profile_set.sh $t "$@" > gpurun_out/${t}_profile_set.log 2>&1; echo "$t: $(tail -c 300 gpurun_out/${t}_bench.json | cut -c1-200)"; }
if [[ $PART == *a* ]]; then
  run ${TAG}
  run ${TAG}_atrium --scene atrium_like
  run ${TAG}_temple --scene temple_like
  run ${TAG}_street --scene street_like
fi
if [[ $PART == *b* ]]; then
  run ${TAG}_cfg0 --scene attenuation_blob --width 512 --height 512 --spp 1 --depth 4 --steps 50 --warmup 5
  run ${TAG}_cfg2 --scene temple_like --spp 64 --depth 8 --in-flight 1 --steps 3 --warmup 2
  run ${TAG}_cfg3 --scene atrium_like --spp 64 --depth 12 --shard 0/4 --in-flight 2 --steps 3 --warmup 2
  run ${TAG}_cfg4 --scene street_like --width 3840 --height 2160 --spp 128 --depth 16 --shard 0/8 --in-flight 2 --steps 3 --warmup 2
fi
