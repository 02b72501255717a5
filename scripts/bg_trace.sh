R=$(pwd); cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_bg
timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_bg -o m -- python $R/bench.py --workload ${1:-bg768} --steps 40 --warmup 5 --no-cpu-baseline --no-secondary > /dev/null 2>&1
DB=$(find /tmp/prof_bg -name '*.db' | head -1)
python $R/scripts/rocpd_stats.py $DB | head -30 | cut -c1-165
python $R/scripts/timeline_busy.py $DB | head -3
