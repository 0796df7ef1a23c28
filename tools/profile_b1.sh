#!/bin/bash
# B = 1 kernel traces of the BASELINE configs as stated (one trajectory per GPU).  Usage: tools/profile_b1.sh <tag>
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
for W in tgv2d rpf2d ldc3d tgv3d; do
  B="python bench.py --no-cpu-baseline --no-other-configs --no-pmc --workload $W --batch 1 --steps 20 --warmup 20 --no-f32 --repeats 3"
  $B > gpurun_out/$TAG/bench_${W}_b1.json 2> gpurun_out/$TAG/bench_${W}_b1.err
  rm -rf /tmp/p_$W
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_$W -- $B > gpurun_out/$TAG/kt_$W.log 2>&1
  python tools/rocpd_summary.py $(find /tmp/p_$W -name "*.db" | head -1) > gpurun_out/$TAG/${TAG}_${W}_b1_kernel_trace.txt 2>&1
done
S="python bench.py --no-cpu-baseline --no-other-configs --no-pmc --model segnn --workload dam2d --batch 1 --steps 20 --warmup 20"
$S > gpurun_out/$TAG/bench_segnn_dam2d_b1.json 2> gpurun_out/$TAG/bench_segnn_dam2d_b1.err
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_sg -- $S > gpurun_out/$TAG/kt_segnn.log 2>&1
python tools/rocpd_summary.py $(find /tmp/p_sg -name "*.db" | head -1) > gpurun_out/$TAG/${TAG}_segnn_dam2d_b1_kernel_trace.txt 2>&1
