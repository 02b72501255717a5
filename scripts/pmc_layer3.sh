# SQ + memory-side counter breakdown of one conv layer in a loop: scripts/pmc_layer3.sh <layer> <batch>   (environment passes through)
R=$(pwd); L=${1:-enc3}; B=${2:-32}
cd /tmp && export TMPDIR=/tmp
i=0
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
            "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INSTS_MFMA GRBM_GUI_ACTIVE" \
            "SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_WAVES" \
            "TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES TCP_TOTAL_CACHE_ACCESSES" \
            "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1)); rm -rf /tmp/pmc_${L}_$i
  timeout 300 rocprofv3 --kernel-trace --pmc $pass -d /tmp/pmc_${L}_$i -o p -- python $R/scripts/conv_microbench.py $L 10 $B > /dev/null 2>&1
done
python $R/scripts/pmc_breakdown.py $(find /tmp/pmc_${L}_1 /tmp/pmc_${L}_2 /tmp/pmc_${L}_3 /tmp/pmc_${L}_4 /tmp/pmc_${L}_5 -name '*.db') 2>&1 | grep -A40 "conv_\|narrow" | head -50
