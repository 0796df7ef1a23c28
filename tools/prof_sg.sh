cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
B="python bench.py --model segnn --workload dam2d --batch 8 --steps 20 --warmup 20 --no-cpu-baseline"
rm -rf /tmp/p_a /tmp/p_b
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -d /tmp/p_a -- $B > /dev/null 2>&1
python tools/rocpd_summary.py $(find /tmp/p_a -name "*.db" | head -1) 2>&1 | grep "k_sg_msg\|k_sg_upd\|PMC" | cut -c1-400
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES -d /tmp/p_b -- $B > /dev/null 2>&1
python tools/rocpd_summary.py $(find /tmp/p_b -name "*.db" | head -1) 2>&1 | grep "k_sg_msg\|k_sg_upd\|PMC" | cut -c1-400
