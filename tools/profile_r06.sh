#!/bin/bash
# Round-6 profile artefacts (run on the GPU box through gpurun; the summaries are copied to profiles/ afterwards).
# Every pass wraps the SAME command with the bench's own --steps / --warmup: kernel trace, PMC traffic and the bench's
# HIP-event timers cover the same rollout steps (the headline workload: stationary TGV3D x 8, vel_amp 0.03).
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06p
mkdir -p $O
python bench.py > $O/r06_bench_tgv3d_b8.json 2> $O/bench_default.err
B="python bench.py --no-cpu-baseline --no-other-configs --no-f32 --repeats 1 --steps 20 --warmup 20"
rm -rf /tmp/p_kt /tmp/p_f /tmp/p_w /tmp/p_sq /tmp/p_sq2
rocprofv3 --kernel-trace --stats -d /tmp/p_kt -- $B > $O/kt.json 2> $O/kt.log
python tools/rocpd_summary.py $(find /tmp/p_kt -name "*.db" | head -1) > $O/r06_kernel_trace_stats.txt 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/p_f -- $B > $O/f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/p_w -- $B > $O/w.log 2>&1
cp profiles/pmc_traffic.json $O/pmc_traffic.json
python tools/pmc_traffic.py tgv3d_b8_st $(find /tmp/p_f -name "*.db" | head -1) $(find /tmp/p_w -name "*.db" | head -1) $O/pmc_traffic.json > $O/r06_pmc_traffic.txt 2>&1
# (pmc_traffic.py merges into the table it finds under profiles/: give it the copy)
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU -d /tmp/p_sq -- $B > $O/sq.log 2>&1
python tools/rocpd_summary.py $(find /tmp/p_sq -name "*.db" | head -1) > $O/r06_pmc_sq.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT -d /tmp/p_sq2 -- $B > $O/sq2.log 2>&1
python tools/rocpd_summary.py $(find /tmp/p_sq2 -name "*.db" | head -1) >> $O/r06_pmc_sq.txt 2>&1
python tools/roofline_check.py $O/kt.json $O/r06_kernel_trace_stats.txt $O/pmc_traffic.json $O/r06_pmc_sq.txt > $O/r06_roofline_check.txt 2>&1
cat $O/r06_roofline_check.txt
# one trajectory per GPU (BASELINE configs as literally stated): kernel traces
for W in tgv2d rpf2d tgv3d ldc3d; do
  rm -rf /tmp/p_b1
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_b1 -- python bench.py --no-cpu-baseline --no-other-configs --workload $W --batch 1 --steps 20 --warmup 20 --no-f32 --repeats 3 > $O/b1_$W.log 2>&1
  python tools/rocpd_summary.py $(find /tmp/p_b1 -name "*.db" | head -1) 2>&1 | head -28 | cut -c1-170 > $O/r06_${W}_b1_kernel_trace.txt
done
rm -rf /tmp/p_sg
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_sg -- python bench.py --model segnn --workload dam2d --batch 1 --steps 20 --warmup 20 --no-cpu-baseline > $O/sg_b1.log 2>&1
python tools/rocpd_summary.py $(find /tmp/p_sg -name "*.db" | head -1) 2>&1 | head -28 | cut -c1-170 > $O/r06_segnn_dam2d_b1_kernel_trace.txt
# the training step (section 8 row N4)
for W in tgv2d tgv3d; do
  rm -rf /tmp/p_tr
  rocprofv3 --kernel-trace --stats -d /tmp/p_tr -- python tools/train_profile.py $W 7 > $O/train_$W.log 2>&1
  python tools/rocpd_summary.py $(find /tmp/p_tr -name "*.db" | head -1) 2>&1 | cut -c1-170 > $O/r06_train_${W}_kernel_trace.txt
  python tools/rocpd_gaps.py $(find /tmp/p_tr -name "*.db" | head -1) >> $O/r06_train_${W}_kernel_trace.txt 2>&1
done
rm -rf /tmp/p_tr
rocprofv3 --kernel-trace --stats -d /tmp/p_tr -- python tools/train_profile.py dam2d 7 segnn > $O/train_segnn.log 2>&1
python tools/rocpd_summary.py $(find /tmp/p_tr -name "*.db" | head -1) 2>&1 | cut -c1-170 > $O/r06_train_segnn_kernel_trace.txt
(python tools/train_profile.py tgv2d 20; python tools/train_profile.py tgv3d 20; python tools/train_profile.py dam2d 20 segnn) 2>&1 | grep -v amdgpu.ids > $O/r06_train_step_ms.txt
