#!/bin/bash
# two-tile node kernel (k_node16s2): parity subset + headline A/B
cd $GRAFT_REPO_ROOT
LB_NODE_T2=2 LB_MSPLIT=0 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "test_gns_forward_parity or (test_fused_rollout_parity and small2d)" 2>&1 | tail -3
for v in 0 1 0 1; do
  LB_NODE_T2=$v python bench.py --no-cpu-baseline --no-other-configs --no-pmc --no-f32 --repeats 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B8 T2=$v', round(d['ms_per_step'],4), round(d['value']/1e6,2), d['breakdown_ms_per_step'])"
done
