R=$(pwd); cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_inf
timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_inf -o m -- python $R/bench.py --workload fg_infer --steps 300 --warmup 10 --no-cpu-baseline --no-secondary > /dev/null 2>&1
DB=$(find /tmp/prof_inf -name '*.db' | head -1)
python $R/scripts/rocpd_stats.py $DB | head -34 | cut -c1-150
python $R/scripts/timeline_busy.py $DB | head -12
python $R/scripts/stream_busy.py $DB | tail -4
