#!/bin/bash
# Round 5, second half: the training-step artefacts after k_lin32h / k_dw_part_h (the training part of tools/profile_r05.sh).
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05t
mkdir -p $O
rm -rf /tmp/p_tr /tmp/p_tr2 /tmp/p_sgt
rocprofv3 --kernel-trace --stats -d /tmp/p_tr -- python tools/train_profile.py tgv3d 7 > $O/train3d.log 2>&1
rocprofv3 --kernel-trace --stats -d /tmp/p_tr2 -- python tools/train_profile.py tgv2d 7 > $O/train2d.log 2>&1
python tools/rocpd_summary.py $(find /tmp/p_tr -name "*.db" | head -1) > $O/r05_train_tgv3d_kernel_trace_h.txt 2>&1
python tools/rocpd_gaps.py $(find /tmp/p_tr -name "*.db" | head -1) >> $O/r05_train_tgv3d_kernel_trace_h.txt 2>&1
python tools/rocpd_summary.py $(find /tmp/p_tr2 -name "*.db" | head -1) > $O/r05_train_tgv2d_kernel_trace_h.txt 2>&1
python tools/rocpd_gaps.py $(find /tmp/p_tr2 -name "*.db" | head -1) >> $O/r05_train_tgv2d_kernel_trace_h.txt 2>&1
rocprofv3 --kernel-trace --stats -d /tmp/p_sgt -- python tools/train_profile.py dam2d 7 segnn > $O/train_segnn.log 2>&1
python tools/rocpd_summary.py $(find /tmp/p_sgt -name "*.db" | head -1) > $O/r05_train_segnn_kernel_trace_h.txt 2>&1
(python tools/train_profile.py tgv3d 20; python tools/train_profile.py tgv2d 20; python tools/train_profile.py dam2d 20 segnn) > $O/r05_train_step_ms_h.txt 2>&1
(LB_TRAIN_MATH=f32 python tools/train_profile.py tgv3d 20; LB_TRAIN_MATH=f32 python tools/train_profile.py tgv2d 20; LB_TRAIN_MATH=f32 python tools/train_profile.py dam2d 20 segnn) > $O/r05_train_step_ms_f32.txt 2>&1
mkdir -p tools/bin
[ -x tools/bin/lin_bench ] && tools/bin/lin_bench 109000 8000 > $O/r05_lin_bench_h.txt 2>&1
tail -3 $O/r05_train_step_ms_h.txt; tail -3 $O/r05_train_step_ms_f32.txt
