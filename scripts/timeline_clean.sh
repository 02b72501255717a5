# kernel trace of a long replayed run: the 55-95 % window of timeline_busy.py then lies inside the timed steps
R=$(pwd); cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_clean
timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_clean -o bench -- python $R/bench.py --steps 300 --warmup 5 --preheat-seconds 0.5 --no-cpu-baseline --no-kernel-events > /dev/null 2>&1
DB=$(find /tmp/prof_clean -name '*.db' | head -1)
python $R/scripts/timeline_busy.py $DB | head -2; python $R/scripts/alone_time.py $DB
