# GPU busy fraction / overlap inside the replayed train step of a block type: busy_probe.sh <Residual|MRU|Pix2Pix> [bench args]
R=$(pwd); BT=$1; shift
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_busy
timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_busy -o r -- python $R/bench.py --block-type $BT --steps 40 --warmup 5 --preheat-seconds 0 --no-cpu-baseline --no-secondary --no-kernel-events --no-gen-fb "$@" > /dev/null 2>&1
DB=$(find /tmp/prof_busy -name '*.db' | head -1)
python $R/scripts/timeline_busy.py $DB | head -14
python $R/scripts/stream_busy.py $DB | tail -5
