# memory-side + SQ counters of the recurrent step alone (scripts/lstm_step_probe.py <rows>): scripts/pmc_lstm.sh [rows]
R=$(pwd); ROWS=${1:-576}
cd /tmp && export TMPDIR=/tmp
i=0
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
            "FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE" \
            "TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES TCP_TOTAL_CACHE_ACCESSES" \
            "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1)); rm -rf /tmp/pmc_lstm_$i
  PYTHONPATH=$R timeout 300 rocprofv3 --kernel-trace --pmc $pass -d /tmp/pmc_lstm_$i -o p -- python $R/scripts/lstm_step_probe.py $ROWS > /dev/null 2>&1
done
python $R/scripts/pmc_breakdown.py $(find /tmp/pmc_lstm_1 /tmp/pmc_lstm_2 /tmp/pmc_lstm_3 /tmp/pmc_lstm_4 -name '*.db') 2>&1 | grep -A32 "lstm_step" | head -120
