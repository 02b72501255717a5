cd /root/repo
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_cli.py -x -q 2>&1 | tail -5
timeout 300 python bench.py --workload fg_mru --steps 10 --warmup 2 > gpurun_out/bench_fg_mru.json 2> gpurun_out/bench_fg_mru.err; tail -c 300 gpurun_out/bench_fg_mru.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d /root/repo/gpurun_out/prof_mru -o mru -- python /root/repo/bench.py --workload fg_mru --steps 3 --warmup 1 > /dev/null 2>&1
python /root/repo/scripts/rocpd_stats.py /root/repo/gpurun_out/prof_mru/mru_results.db > /root/repo/gpurun_out/mru_kernel_stats.txt 2>&1
