# PMC passes of the headline step (separate passes, kernel trace only; never TA_*/TCP_* counters: they hang on this pool) + the
# bench line they are matched with.  usage: run_profile_pmc.sh <round tag, e.g. r03>.  Run through gpurun from the repo root; the
# summary lands in gpurun_out/<tag>_pmc.json (copy to profiles/: bench.py attaches it when the kernel-source hash matches).
R=$(pwd); TAG=${1:-r03}
mkdir -p gpurun_out
timeout 900 python bench.py --no-cpu-baseline --no-secondary > gpurun_out/bench_for_pmc.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 2 --warmup 1 --preheat-seconds 0 --no-graphs --no-kernel-events --no-cpu-baseline --no-secondary"
rm -rf /tmp/pmcflat
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_VALU"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rm -rf /tmp/pmc_final_$tag
  timeout 400 rocprofv3 --kernel-trace --pmc $pass -d /tmp/pmc_final_$tag -o p -- $CMD > /dev/null 2>&1
  echo "$tag rc=$?"
  mkdir -p /tmp/pmcflat/pmc_final_$tag; cp $(find /tmp/pmc_final_$tag -name '*.db' | head -1) /tmp/pmcflat/pmc_final_$tag/p_results.db
done
# kernel trace of the graph-REPLAYED step (no counters): per-kernel time inside the step
rm -rf /tmp/trace_replay
timeout 400 rocprofv3 --kernel-trace -d /tmp/trace_replay -o p -- python $R/bench.py --steps 30 --warmup 5 --preheat-seconds 0 --no-kernel-events --no-cpu-baseline --no-secondary --no-gen-fb > /dev/null 2>&1
echo "trace_replay rc=$?"
mkdir -p /tmp/pmcflat/trace_replay; cp $(find /tmp/trace_replay -name '*.db' | head -1) /tmp/pmcflat/trace_replay/p_results.db
python $R/scripts/rocpd_stats.py /tmp/pmcflat/trace_replay/p_results.db > $R/gpurun_out/${TAG}_bench_n1_kernel_stats.txt 2>&1
cd $R
python scripts/pmc_summary.py /tmp/pmcflat gpurun_out/${TAG}_pmc.json gpurun_out/bench_for_pmc.json | tail -16
