#!/bin/bash
# SQ counters of the neighbor-search kernel, wave-per-cell (k_nlc) vs wave-per-receiver (k_nlw): tools/prof_nl.sh [tag]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
TAG=${1:-x}; shift; O=gpurun_out/nl_$TAG
mkdir -p $O
B="python bench.py --no-cpu-baseline --no-other-configs --no-pmc --no-f32 --repeats 1 --steps 20 --warmup 20 $@"
for V in ${NLV:-nlc wave}; do
  export LB_NL_KERNEL=$V
  [ $V = nlc ] && unset LB_NL_KERNEL
  rm -rf /tmp/p_a /tmp/p_b /tmp/p_k
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_k -- $B > $O/kt_$V.log 2>&1
  python tools/rocpd_summary.py $(find /tmp/p_k -name "*.db" | head -1) 2>&1 | grep -i 'k_nl\|k_cell\|k_scan\|k_row\|calls' | cut -c1-170 > $O/kt_$V.txt
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -d /tmp/p_a -- $B > $O/a_$V.log 2>&1
  python tools/rocpd_summary.py $(find /tmp/p_a -name "*.db" | head -1) 2>&1 | grep "k_nlc\|k_nlw\|PMC" > $O/sq_$V.txt
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA -d /tmp/p_b -- $B > $O/b_$V.log 2>&1
  python tools/rocpd_summary.py $(find /tmp/p_b -name "*.db" | head -1) 2>&1 | grep "k_nlc\|k_nlw\|PMC" >> $O/sq_$V.txt
done
tail -n 40 $O/kt_*.txt $O/sq_*.txt
