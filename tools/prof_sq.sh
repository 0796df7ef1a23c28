cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU -d /tmp/p_sq -- python bench.py --no-cpu-baseline --steps 5 --warmup 5 > gpurun_out/p_sq.log 2>&1
python tools/rocpd_summary.py $(find /tmp/p_sq -name "*.db" | head -1) > gpurun_out/r02_pmc_sq.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT -d /tmp/p_sq2 -- python bench.py --no-cpu-baseline --steps 5 --warmup 5 > gpurun_out/p_sq2.log 2>&1
python tools/rocpd_summary.py $(find /tmp/p_sq2 -name "*.db" | head -1) > gpurun_out/r02_pmc_sq2.txt 2>&1
