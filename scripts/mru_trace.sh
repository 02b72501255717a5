R=$(pwd); cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_mru
timeout 900 rocprofv3 --kernel-trace -d /tmp/prof_mru -o m -- python $R/bench.py --block-type MRU --steps 6 --warmup 3 --preheat-seconds 0 --no-cpu-baseline --no-kernel-events > /dev/null 2>&1
DB=$(find /tmp/prof_mru -name '*.db' | head -1)
python $R/scripts/rocpd_stats.py $DB | head -60 | cut -c1-175
