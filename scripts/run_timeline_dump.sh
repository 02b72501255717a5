# gpurun_out/timeline_<tag>.txt: the dispatches of one replayed train iteration (scripts/timeline_dump.py)
R=$(pwd); TAG=${1:-a}
mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_tl
timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_tl -o bench -- python $R/bench.py --steps 12 --warmup 5 --preheat-seconds 0.5 --no-cpu-baseline --no-secondary --no-kernel-events > /dev/null 2>&1
python $R/scripts/timeline_dump.py $(find /tmp/prof_tl -name '*.db' | head -1) 4 > $R/gpurun_out/timeline_$TAG.txt 2>&1
cd $R; head -3 gpurun_out/timeline_$TAG.txt; wc -l gpurun_out/timeline_$TAG.txt
