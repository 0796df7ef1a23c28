#!/bin/bash
# edge32 ablation: B = 8 headline with parts of k_edge32 removed (1 no psr gathers, 8 no GEMMs, 9 both)
cd $GRAFT_REPO_ROOT
for A in 0 1 8 9; do
  LB_EDGE32=1 LB_E32_ABL=$A python bench.py --no-cpu-baseline --no-other-configs --no-pmc --no-f32 --repeats 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('abl $A', round(d['ms_per_step'],3), d['breakdown_ms_per_step'].get('edge_mlp'))"
done
