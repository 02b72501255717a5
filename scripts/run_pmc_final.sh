# PMC passes on the final round-1 tree (separate passes, kernel-trace only; never TA_*/TCP_* counters: they hang here)
cd /tmp && export TMPDIR=/tmp
CMD="python /root/repo/bench.py --steps 2 --warmup 1 --preheat-seconds 0 --no-graphs --no-kernel-events --no-cpu-baseline"
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_VALU"; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout 400 rocprofv3 --kernel-trace --pmc $pass -d /root/repo/gpurun_out/pmc_final_$tag -o p -- $CMD > /dev/null 2>&1
  echo "$tag rc=$?"
  ls /root/repo/gpurun_out/pmc_final_$tag | head -3
done
