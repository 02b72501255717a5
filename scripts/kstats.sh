# usage: kstats.sh <out-name> <bench args...> : rocprofv3 kernel trace of one bench invocation -> gpurun_out/<out-name>.txt (per-kernel totals / averages)
R=$(pwd); N=$1; shift; cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_$N
timeout 900 rocprofv3 --kernel-trace -d /tmp/prof_$N -o k -- python $R/bench.py --no-cpu-baseline --no-secondary "$@" 2>&1 | grep -v amdgpu | tail -1 | cut -c1-200
DB=$(find /tmp/prof_$N -name '*.db' | head -1)
python $R/scripts/rocpd_stats.py $DB > $R/gpurun_out/$N.txt; head -60 $R/gpurun_out/$N.txt | cut -c1-170; if [ -n "$KSTATS_PROBE" ]; then python $R/scripts/$KSTATS_PROBE $DB > $R/gpurun_out/${N}_probe.txt; fi
