cd /root/repo
mkdir -p gpurun_out
for w in fg_infer fg_resid bg768; do timeout 300 python bench.py --workload $w --steps 10 --warmup 2 > gpurun_out/bench_$w.json 2> gpurun_out/bench_$w.err; tail -c 300 gpurun_out/bench_$w.err; done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d /root/repo/gpurun_out/prof_bg -o bg -- python /root/repo/bench.py --workload bg768 --steps 3 --warmup 1 > /dev/null 2>&1
python /root/repo/scripts/rocpd_stats.py /root/repo/gpurun_out/prof_bg/bg_results.db > /root/repo/gpurun_out/bg_kernel_stats.txt 2>&1
