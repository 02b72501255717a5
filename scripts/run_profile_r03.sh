# Round-3 kernel-trace summary of the default bench command (hipGraph replay): gpurun_out/r03_bench_n1_kernel_stats_<tag>.txt
R=$(pwd); TAG=${1:-a}
mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_r03
timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_r03 -o bench -- python $R/bench.py --steps 20 --warmup 5 --preheat-seconds 0.5 --no-cpu-baseline --no-secondary --no-kernel-events > $R/gpurun_out/bench_under_rocprof_r03_$TAG.json 2>/dev/null
python $R/scripts/rocpd_stats.py $(find /tmp/prof_r03 -name '*.db' | head -1) > $R/gpurun_out/r03_bench_n1_kernel_stats_$TAG.txt 2>&1
python $R/scripts/timeline_busy.py $(find /tmp/prof_r03 -name '*.db' | head -1) >> $R/gpurun_out/r03_bench_n1_kernel_stats_$TAG.txt 2>&1
python $R/scripts/stream_busy.py $(find /tmp/prof_r03 -name "*.db" | head -1) >> $R/gpurun_out/r03_bench_n1_kernel_stats_$TAG.txt 2>&1
cd $R; head -24 gpurun_out/r03_bench_n1_kernel_stats_$TAG.txt
