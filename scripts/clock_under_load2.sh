#!/bin/bash
# shader clock seen from inside a kernel while (a) nothing, (b) our conv, (c) the vendor sgemm runs in another process
B=sketchyscenecolorization_amd/lib/clock_probe_bench
echo idle; $B 2
for m in conv mm; do
  python /tmp/load_conv.py $m & pid=$!
  sleep 5; echo "under $m"; $B 4; rocm-smi --showpower 2>/dev/null | grep -i "power (W)"
  wait $pid
done
