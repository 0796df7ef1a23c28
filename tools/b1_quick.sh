#!/bin/bash
# quick B = 1 bench lines (no traces): tools/b1_quick.sh [workloads...]
cd $GRAFT_REPO_ROOT
for W in ${@:-tgv2d rpf2d ldc3d tgv3d}; do
  python bench.py --no-cpu-baseline --no-other-configs --no-pmc --workload $W --batch 1 --steps 20 --warmup 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$W', round(d['ms_per_step'],4), round(d['value']/1e6,2), d.get('breakdown_ms_per_step'))"
done
