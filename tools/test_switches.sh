#!/bin/bash
# The parity subset (forward / rollout / determinism / overflow tests of tests/test_gpu_parity.py) under every
# ablation switch that selects a different kernel or schedule: the kept round-1 kernels and the measured-and-rejected
# variants must stay correct.  Run on the GPU box:  gpurun -- bash tools/test_switches.sh
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
for e in "LB_MATH=f32" "LB_EDGE_KERNEL=n" "LB_EDGE_KERNEL=v6" "LB_EDGE_KERNEL=v3" "LB_NODE_KERNEL=h LB_NODE_LOADERS=0" \
         "LB_ENC_KERNEL=h LB_DEC_KERNEL=h" "LB_FUSED_AGG=0" "LB_SMALL_FUSED=0 LB_NL_KERNEL=wave" \
         "LB_NL_KERNEL=cell LB_NODE_S_MIN=1" "LB_EDGE_TILE=32" "LB_GRAPH=1"; do
  echo "== $e"
  env $e python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "forward or rollout or bitwise or overflow" 2>&1 | tail -2
done
