#!/bin/bash
# kernel trace of one B = 1 workload: tools/kt_one.sh <workload> [extra bench args]  -> gpurun_out/kt_<workload>.txt
W=${1:-tgv2d}; shift
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
B="python bench.py --no-cpu-baseline --no-other-configs --no-pmc --workload $W --batch 1 --steps 20 --warmup 20 --no-f32 --repeats 3 $@"
rm -rf /tmp/p_$W
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_$W -- $B > gpurun_out/kt_$W.log 2>&1
python tools/rocpd_summary.py $(find /tmp/p_$W -name "*.db" | head -1) > gpurun_out/kt_$W.txt 2>&1
head -24 gpurun_out/kt_$W.txt | cut -c1-160
