#!/bin/bash
# Shader clock (scripts/clock_probe.hip, a second process) while the replayed train iteration runs in another process.
B=sketchyscenecolorization_amd/lib/clock_probe_bench
[ -x $B ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 scripts/clock_probe.hip -o $B
echo idle; $B 2
python scripts/power_trace.py 14 > /tmp/pt.log 2>&1 & pid=$!
sleep 9; echo "under the train step ($*)"; $B 3; rocm-smi --showpower --showclocks 2>/dev/null | grep -i "power\|sclk" | head -4
wait $pid; tail -1 /tmp/pt.log
