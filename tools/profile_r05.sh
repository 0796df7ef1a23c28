#!/bin/bash
# Round-5 profile artefacts (run on the GPU box through gpurun; summaries are copied to profiles/ afterwards).
# Every pass wraps the SAME command with the bench's own --steps / --warmup: kernel trace, PMC traffic
# and the bench's HIP-event timers then cover the same rollout steps, i.e. the same mean edge count.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05p
mkdir -p $O
python bench.py > $O/r05_bench_tgv3d_b8.json 2> $O/bench_default.err
B="python bench.py --no-cpu-baseline --no-other-configs --no-pmc --no-f32 --repeats 1 --steps 20 --warmup 20"
rm -rf /tmp/p_kt /tmp/p_f /tmp/p_w /tmp/p_sq /tmp/p_sq2
rocprofv3 --kernel-trace --stats -d /tmp/p_kt -- $B > $O/kt.json 2> $O/kt.log
python tools/rocpd_summary.py $(find /tmp/p_kt -name "*.db" | head -1) > $O/r05_kernel_trace_stats.txt 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/p_f -- $B > $O/f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/p_w -- $B > $O/w.log 2>&1
python tools/pmc_traffic.py tgv3d_b8 $(find /tmp/p_f -name "*.db" | head -1) $(find /tmp/p_w -name "*.db" | head -1) $O/pmc_traffic.json > $O/r05_pmc_traffic.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU -d /tmp/p_sq -- $B > $O/sq.log 2>&1
python tools/rocpd_summary.py $(find /tmp/p_sq -name "*.db" | head -1) > $O/r05_pmc_sq.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT -d /tmp/p_sq2 -- $B > $O/sq2.log 2>&1
python tools/rocpd_summary.py $(find /tmp/p_sq2 -name "*.db" | head -1) >> $O/r05_pmc_sq.txt 2>&1
# the roofline of the bench line recomputed from these files alone + the MFMA-per-tile self-check
python tools/roofline_check.py $O/kt.json $O/r05_kernel_trace_stats.txt $O/pmc_traffic.json $O/r05_pmc_sq.txt > $O/r05_roofline_check.txt 2>&1
cat $O/r05_roofline_check.txt
# the training step (section 8 row N4): kernel traces of both workloads + dispatch-gap summary, the GEMM micro-benchmark
rm -rf /tmp/p_tr /tmp/p_tr2
rocprofv3 --kernel-trace --stats -d /tmp/p_tr -- python tools/train_profile.py tgv3d 7 > $O/train3d.log 2>&1
rocprofv3 --kernel-trace --stats -d /tmp/p_tr2 -- python tools/train_profile.py tgv2d 7 > $O/train2d.log 2>&1
python tools/rocpd_summary.py $(find /tmp/p_tr -name "*.db" | head -1) > $O/r05_train_tgv3d_kernel_trace.txt 2>&1
python tools/rocpd_gaps.py $(find /tmp/p_tr -name "*.db" | head -1) >> $O/r05_train_tgv3d_kernel_trace.txt 2>&1
python tools/rocpd_summary.py $(find /tmp/p_tr2 -name "*.db" | head -1) > $O/r05_train_tgv2d_kernel_trace.txt 2>&1
python tools/rocpd_gaps.py $(find /tmp/p_tr2 -name "*.db" | head -1) >> $O/r05_train_tgv2d_kernel_trace.txt 2>&1
(python tools/train_profile.py tgv3d 20; python tools/train_profile.py tgv2d 20) > $O/r05_train_step_ms.txt 2>&1
[ -x tools/bin/lin_bench ] && tools/bin/lin_bench 109000 8000 > $O/r05_lin_bench.txt 2>&1
# SEGNN training step (config 5's model, lb_train_segnn.h)
rm -rf /tmp/p_sgt
rocprofv3 --kernel-trace --stats -d /tmp/p_sgt -- python tools/train_profile.py dam2d 7 segnn > $O/train_segnn.log 2>&1
python tools/rocpd_summary.py $(find /tmp/p_sgt -name "*.db" | head -1) > $O/r05_train_segnn_kernel_trace.txt 2>&1
python tools/rocpd_gaps.py $(find /tmp/p_sgt -name "*.db" | head -1) >> $O/r05_train_segnn_kernel_trace.txt 2>&1
python tools/train_profile.py dam2d 20 segnn >> $O/r05_train_step_ms.txt 2>&1
