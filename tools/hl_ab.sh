#!/bin/bash
# headline A/B on one box: tools/hl_ab.sh "<ENV_A>" "<ENV_B>"   (e.g. "LB_STEP_FUSE=1" "LB_STEP_FUSE=0")
cd $GRAFT_REPO_ROOT
for i in 1 2; do
  for E in "$1" "$2"; do
    env $E python bench.py --no-cpu-baseline --no-other-configs --no-pmc --no-f32 --repeats 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$E', round(d['ms_per_step'],4), round(d['value']/1e6,2), d['breakdown_ms_per_step'])"
  done
done
