R=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_resid_train
timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_resid_train -o r -- python $R/bench.py --block-type Residual --steps 3 --warmup 3 --preheat-seconds 0 --no-cpu-baseline --no-kernel-events > /dev/null 2>&1
python $R/scripts/rocpd_stats.py $R/gpurun_out/prof_resid_train/r_results.db > $R/gpurun_out/resid_train_kernel_stats.txt 2>&1
rm -rf $R/gpurun_out/prof_resid_train
