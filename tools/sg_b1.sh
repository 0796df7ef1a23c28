#!/bin/bash
# SEGNN DAM2D B = 1 quick line + neighbor kernels of its trace
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
S="python bench.py --no-cpu-baseline --no-other-configs --no-pmc --model segnn --workload dam2d --batch 1 --steps 20 --warmup 20 --no-f32 --repeats 3"
$S 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('segnn dam2d b1', round(d['ms_per_step'],4), d.get('breakdown_ms_per_step'))"
rm -rf /tmp/p_sg; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_sg -- $S > /dev/null 2>&1
python tools/rocpd_summary.py $(find /tmp/p_sg -name "*.db" | head -1) | grep -E "k_nl|k_node_feat|k_sg_node_prep" | cut -c1-150
