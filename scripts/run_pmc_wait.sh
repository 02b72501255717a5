cd /tmp && export TMPDIR=/tmp
CMD="python /root/repo/bench.py --steps 2 --warmup 1 --preheat-seconds 0 --no-graphs --no-kernel-events --no-cpu-baseline"
timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d /root/repo/gpurun_out/pmc_final_WAIT -o p -- $CMD > /dev/null 2>&1; echo rc=$?
