cd /root/repo
mkdir -p gpurun_out
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_torchrun_n1.json 2> gpurun_out/bench_torchrun_n1.err
tail -c 300 gpurun_out/bench_torchrun_n1.err
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d /root/repo/gpurun_out/prof_r1d -o bench -- python /root/repo/bench.py --steps 10 --warmup 3 --no-cpu-baseline > /root/repo/gpurun_out/bench_under_rocprof.json 2>/dev/null
python /root/repo/scripts/rocpd_stats.py /root/repo/gpurun_out/prof_r1d/bench_results.db > /root/repo/gpurun_out/r01_bench_n1_kernel_stats_v3.txt 2>&1
