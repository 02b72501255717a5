R=$(pwd); cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_gfb
timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_gfb -o g -- python $R/scripts/gen_fb_trace.py 2>&1 | grep -v amdgpu | tail -2
DB=$(find /tmp/prof_gfb -name '*.db' | head -1)
python $R/scripts/rocpd_stats.py $DB | head -45 | cut -c1-150
python $R/scripts/gfb_timeline.py $DB > $R/gpurun_out/gfb_timeline.txt; tail -1 $R/gpurun_out/gfb_timeline.txt
