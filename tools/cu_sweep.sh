#!/bin/bash
# How many CUs does the HBM-bound processor edge kernel need?    TGV3D-8k x 8.
cd $GRAFT_REPO_ROOT
for G in 256 224 192 160 128; do
  LB_EDGE_GRID=$G python bench.py --no-cpu-baseline --no-other-configs --no-pmc --no-f32 --repeats 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('edge grid $G: ms/step', round(d['ms_per_step'],4), 'edge us/launch', round(d['roofline']['us_per_launch'],1), 'frac', round(d['roofline']['frac'],3))"
done
