R=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_bg_train
timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_bg_train -o r -- python $R/bench.py --workload bg768_train --steps 5 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
python $R/scripts/rocpd_stats.py $R/gpurun_out/prof_bg_train/r_results.db > $R/gpurun_out/bg_train_kernel_stats.txt 2>&1
rm -rf $R/gpurun_out/prof_bg_train
