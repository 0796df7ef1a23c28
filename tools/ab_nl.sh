#!/bin/bash
# A/B of the neighbor-search kernels on the headline workload: tools/ab_nl.sh  -> prints value / ms_per_step / breakdown per variant
cd $GRAFT_REPO_ROOT
for V in nlc wave; do
  export LB_NL_KERNEL=$V; [ $V = nlc ] && unset LB_NL_KERNEL
  python bench.py --no-cpu-baseline --no-other-configs --no-pmc --no-f32 "$@" > gpurun_out/ab_$V.json 2>gpurun_out/ab_$V.err
  python -c "
import json,sys
d=json.loads(open('gpurun_out/ab_$V.json').read().strip().splitlines()[-1]); print('$V', round(d['value']), round(d['ms_per_step'],4), d.get('breakdown_ms_per_step'))"
done
