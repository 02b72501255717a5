cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d /root/repo/gpurun_out/prof_mru_train -o mru -- python /root/repo/bench.py --block-type MRU --steps 3 --warmup 3 --preheat-seconds 0 --no-cpu-baseline --no-kernel-events > /dev/null 2>&1
python /root/repo/scripts/rocpd_stats.py /root/repo/gpurun_out/prof_mru_train/mru_results.db > /root/repo/gpurun_out/mru_train_kernel_stats.txt 2>&1
