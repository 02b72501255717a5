#!/bin/bash
# Shader clock seen from inside a kernel (scripts/clock_probe.hip) and socket power while (a) nothing, (b) our conv
# kernel, (c) the vendor sgemm runs in another process.  Build the probe first:
#   hipcc --offload-arch=gfx950 -O3 scripts/clock_probe.hip -o sketchyscenecolorization_amd/lib/clock_probe_bench
B=sketchyscenecolorization_amd/lib/clock_probe_bench
cat > /tmp/load_conv.py <<'PY'
import sys, time, torch
sys.path.insert(0, '.')
from sketchyscenecolorization_amd import hip
from sketchyscenecolorization_amd.hip import View
mode = sys.argv[1]
if mode == 'conv':
    x = torch.randn(16, 64, 64, 1024, device='cuda'); w = torch.randn(1, 1, 1024, 1024, device='cuda') * 0.05
    out = torch.empty(16, 64, 64, 1024, device='cuda'); v = View(x)
    fn = lambda: hip.conv_forward(v, w, 1, 0, out)
else:
    a = torch.randn(65536, 1024, device='cuda'); b = torch.randn(1024, 1024, device='cuda')
    fn = lambda: torch.mm(a, b)
fl = 2.0 * 65536 * 1024 * 1024
t0 = time.time(); n = 0
while time.time() - t0 < 8:
    for _ in range(200): fn()
    torch.cuda.synchronize(); n += 200
print(mode, 'TFLOP/s', fl * n / (time.time() - t0) / 1e12)
PY
echo idle; $B 2
for m in conv mm; do
  python /tmp/load_conv.py $m & pid=$!
  sleep 5; echo "under $m"; $B 4; rocm-smi --showpower 2>/dev/null | grep -i "power (W)"
  wait $pid
done
