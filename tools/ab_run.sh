#!/bin/bash
# tools/ab_run.sh "SCENES" ROUNDS NAME... -- A/B on ONE GPU box: bench.py (headline shape, no children, no CPU leg) for every
# experimental build tools/ab/libptx_NAME.so ("product" = the shipped library), the builds taking turns ROUNDS times.
cd "$(dirname "$0")/.."
SCENES=$1; ROUNDS=$2; shift 2
for r in $(seq 1 $ROUNDS); do
  for n in "$@"; do
    for s in $SCENES; do
      if [ "$n" = product ]; then unset PTX_HIP_LIB; else export PTX_HIP_LIB=$PWD/tools/ab/libptx_$n.so; fi
      python3 bench.py --no-extra-scenes --no-cpu-baseline --scene $s --steps 20 --warmup 3 ${AB_ARGS} > /dev/null 2>&1; python3 -c "
import json,sys
d=json.load(open('bench_detail.json'))   # the full record of the run that has just ended
rf=d.get('roofline',{})
k=d.get('one_in_flight_kernel_ms_per_step',{})
print('$n', '$s', 'round $r', 'value %.1f' % d['value'], 'ms/step %.3f' % d['ms_per_step'], 'alone ms/step %.3f' % d.get('one_in_flight_ms_per_step',0), 'closest launch %.4f' % rf.get('avg_launch_ms',0), 'shade alone %.4f' % rf.get('shade',{}).get('ms_alone',0), 'alone per step: closest %.2f shade %.2f shadow %.2f tail %.2f' % (k.get('k_trace_closest',0), k.get('k_shade',0), k.get('k_trace_shadow',0), k.get('k_tail',0)))"
    done
  done
done
