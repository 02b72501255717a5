"""Where do a wave's cycles go in a K step of wgrad128?  Needs the diagnostic build (scripts/wg128_timing.sh)."""
import ctypes
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sketchyscenecolorization_amd import hip
from sketchyscenecolorization_amd.hip import View
N = 32
g = torch.Generator(device='cuda').manual_seed(0)
r = lambda *s: torch.randn(*s, device='cuda', generator=g)
x, dy, dw = r(N, 48, 48, 128), r(N, 24, 24, 256), torch.empty(4, 4, 128, 256, device='cuda')
fn = lambda: hip.conv_wgrad(View(x, None, None, 2), View(dy), dw, 2, 1)
for _ in range(3):
    fn()
torch.cuda.synchronize()
L = hip.lib()
L.ssc_wg128_timing.argtypes = [ctypes.c_void_p, ctypes.c_int]
L.ssc_wg128_timing(None, 1)
iters = 20
for _ in range(iters):
    fn()
torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 8)()
L.ssc_wg128_timing(out, 0)
k = out[4]
names = ['barrier release -> first operands', 'MFMA groups (incl. staging)', 'counted wait (vmcnt / lgkmcnt)', 'barrier']
tot = sum(out[i] for i in range(4))
for i in range(4):
    print('%-34s %8.1f cycles per K step  %5.1f %%' % (names[i], out[i] / k, 100.0 * out[i] / tot))
print('K steps %d, per K step %.1f cycles (MFMA alone: %d); whole kernel per workgroup %.0f cycles = %.1f per K step'
      % (k, tot / k, 64 * 64, out[5] / (k / 36.0), out[5] / k))
