# Round-1 final tree: default bench line, rocprofv3 kernel-trace summary, PMC passes (separate, kernel-trace only; never
# TA_*/TCP_* counters: they hang on this pool).  Run through gpurun from the repo root; results land in gpurun_out/.
R=$(pwd)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 2 --warmup 1 --preheat-seconds 0 --no-graphs --no-kernel-events --no-cpu-baseline"
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_VALU"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rm -rf $R/gpurun_out/pmc_final_$tag
  timeout 400 rocprofv3 --kernel-trace --pmc $pass -d $R/gpurun_out/pmc_final_$tag -o p -- $CMD > /dev/null 2>&1
  echo "$tag rc=$?"
done
python $R/scripts/pmc_summary.py $R/gpurun_out $R/gpurun_out/r01_pmc_final.json | tail -5
cp $R/gpurun_out/r01_pmc_final.json $R/profiles/r01_pmc_final.json      # bench.py reads the traffic figures from here
rm -rf $R/gpurun_out/prof_v4
timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_v4 -o bench -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/bench_under_rocprof.json 2>/dev/null
python $R/scripts/rocpd_stats.py $R/gpurun_out/prof_v4/bench_results.db > $R/gpurun_out/r01_bench_n1_kernel_stats_v4.txt 2>&1
rm -rf $R/gpurun_out/prof_v4 $R/gpurun_out/pmc_final_*/*.db 2>/dev/null
cd $R
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
tail -c 400 gpurun_out/bench_default.json
