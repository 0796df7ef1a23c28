#!/bin/bash
# Round-4 SEGNN profile artefacts (config 5: DAM2D SEGNN-10-64), run on the GPU box through gpurun:
#   bench lines B = 1 / 8, kernel traces B = 1 / 8, PMC traffic (FETCH_SIZE, WRITE_SIZE) and SQ instruction counters B = 8.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
for B in 1 8; do
  S="python bench.py --no-cpu-baseline --no-other-configs --no-pmc --model segnn --workload dam2d --batch $B --steps 20 --warmup 20"
  $S 2>/dev/null | tail -1 > $O/r04_segnn_dam2d_b$B.json
  rm -rf /tmp/p_sg$B
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_sg$B -- $S > $O/kt_sg$B.log 2>&1
  python tools/rocpd_summary.py $(find /tmp/p_sg$B -name "*.db" | head -1) > $O/r04_segnn_dam2d_b${B}_kernel_trace.txt 2>&1
done
S="python bench.py --no-cpu-baseline --no-other-configs --no-pmc --model segnn --workload dam2d --batch 8 --steps 10 --warmup 5"
rm -rf /tmp/p_sf /tmp/p_sw /tmp/p_sq /tmp/p_sq2
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/p_sf -- $S > $O/sf.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/p_sw -- $S > $O/sw.log 2>&1
python tools/pmc_traffic.py segnn_dam2d_b8 $(find /tmp/p_sf -name "*.db" | head -1) $(find /tmp/p_sw -name "*.db" | head -1) $O/pmc_traffic_segnn.json > $O/r04_segnn_pmc_traffic.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU -d /tmp/p_sq -- $S > $O/sq.log 2>&1
python tools/rocpd_summary.py $(find /tmp/p_sq -name "*.db" | head -1) > $O/r04_segnn_pmc_sq.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_TRANS -d /tmp/p_sq2 -- $S > $O/sq2.log 2>&1
python tools/rocpd_summary.py $(find /tmp/p_sq2 -name "*.db" | head -1) > $O/r04_segnn_pmc_sq2.txt 2>&1
head -30 $O/r04_segnn_dam2d_b1_kernel_trace.txt | cut -c1-150
head -14 $O/r04_segnn_dam2d_b8_kernel_trace.txt | cut -c1-150
head -12 $O/r04_segnn_pmc_traffic.txt | cut -c1-200
head -12 $O/r04_segnn_pmc_sq.txt | cut -c1-250
head -12 $O/r04_segnn_pmc_sq2.txt | cut -c1-250
