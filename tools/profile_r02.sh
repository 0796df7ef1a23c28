#!/bin/bash
# Round-2 profile artefacts (run on the GPU box through gpurun; summaries are copied to profiles/ afterwards):
#   kernel trace + stats, separate PMC passes (FETCH_SIZE / WRITE_SIZE / SQ+MFMA), for the GNS bench
#   (TGV3D-8k x 8) and the SEGNN bench (DAM2D x 8).
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
B="python bench.py --no-cpu-baseline --no-other-configs --no-pmc --steps 10 --warmup 10"
rocprofv3 --kernel-trace --stats -d /tmp/p_kt -- $B > gpurun_out/r02/kt.log 2>&1
python tools/rocpd_summary.py $(find /tmp/p_kt -name "*.db" | head -1) > gpurun_out/r02/r02_kernel_trace_stats.txt 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/p_f -- $B > gpurun_out/r02/f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/p_w -- $B > gpurun_out/r02/w.log 2>&1
python tools/pmc_traffic.py tgv3d_b8 $(find /tmp/p_f -name "*.db" | head -1) $(find /tmp/p_w -name "*.db" | head -1) gpurun_out/r02/pmc_traffic.json > gpurun_out/r02/r02_pmc_traffic.txt 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU -d /tmp/p_sq -- $B > gpurun_out/r02/sq.log 2>&1
python tools/rocpd_summary.py $(find /tmp/p_sq -name "*.db" | head -1) > gpurun_out/r02/r02_pmc_sq.txt 2>&1
S="python bench.py --no-cpu-baseline --no-other-configs --no-pmc --model segnn --workload dam2d --steps 10 --warmup 10"
rocprofv3 --kernel-trace --stats -d /tmp/p_skt -- $S > gpurun_out/r02/skt.log 2>&1
python tools/rocpd_summary.py $(find /tmp/p_skt -name "*.db" | head -1) > gpurun_out/r02/r02_segnn_kernel_trace_stats.txt 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/p_sf -- $S > gpurun_out/r02/sf.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/p_sw -- $S > gpurun_out/r02/sw.log 2>&1
python tools/pmc_traffic.py segnn_dam2d_b8 $(find /tmp/p_sf -name "*.db" | head -1) $(find /tmp/p_sw -name "*.db" | head -1) gpurun_out/r02/pmc_traffic_segnn.json > gpurun_out/r02/r02_segnn_pmc_traffic.txt 2>&1
python bench.py > gpurun_out/r02/bench_default.json 2> gpurun_out/r02/bench_default.err
python bench.py --no-cpu-baseline --no-other-configs --model segnn --workload dam2d --batch 1 > gpurun_out/r02/bench_segnn_dam2d_b1.json 2>/dev/null
python bench.py --no-cpu-baseline --no-other-configs --model segnn --workload dam2d --batch 8 > gpurun_out/r02/bench_segnn_dam2d_b8.json 2>/dev/null
