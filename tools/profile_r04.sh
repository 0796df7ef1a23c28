#!/bin/bash
# Round-4 profile artefacts (run on the GPU box through gpurun; summaries are copied to profiles/ afterwards):
#   default bench line, kernel trace of the headline config, B = 1 kernel traces of every BASELINE config
#   (tools/profile_b1.sh), separate PMC passes (FETCH_SIZE / WRITE_SIZE) of the headline config.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
python bench.py > gpurun_out/r04/r04_bench_tgv3d_b8.json 2> gpurun_out/r04/bench_default.err
B="python bench.py --no-cpu-baseline --no-other-configs --no-pmc --no-f32 --repeats 1 --steps 10 --warmup 10"
rm -rf /tmp/p_kt /tmp/p_f /tmp/p_w
rocprofv3 --kernel-trace --stats -d /tmp/p_kt -- $B > gpurun_out/r04/kt.log 2>&1
python tools/rocpd_summary.py $(find /tmp/p_kt -name "*.db" | head -1) > gpurun_out/r04/r04_kernel_trace_stats.txt 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/p_f -- $B > gpurun_out/r04/f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/p_w -- $B > gpurun_out/r04/w.log 2>&1
python tools/pmc_traffic.py tgv3d_b8 $(find /tmp/p_f -name "*.db" | head -1) $(find /tmp/p_w -name "*.db" | head -1) gpurun_out/r04/pmc_traffic.json > gpurun_out/r04/r04_pmc_traffic.txt 2>&1
# SQ instruction counters of the headline config (VALU / LDS instructions per tile), two passes
rm -rf /tmp/p_sq /tmp/p_sq2
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU -d /tmp/p_sq -- $B > gpurun_out/r04/sq.log 2>&1
python tools/rocpd_summary.py $(find /tmp/p_sq -name "*.db" | head -1) > gpurun_out/r04/r04_pmc_sq.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_TRANS -d /tmp/p_sq2 -- $B > gpurun_out/r04/sq2.log 2>&1
python tools/rocpd_summary.py $(find /tmp/p_sq2 -name "*.db" | head -1) > gpurun_out/r04/r04_pmc_sq2.txt 2>&1
# training step (row N4): kernel trace
rm -rf /tmp/p_tr
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_tr -- python tools/train_profile.py tgv3d 5 > gpurun_out/r04/train_kt.log 2>&1
python tools/rocpd_summary.py $(find /tmp/p_tr -name "*.db" | head -1) > gpurun_out/r04/r04_train_tgv3d_kernel_trace.txt 2>&1
bash tools/profile_b1.sh r04
