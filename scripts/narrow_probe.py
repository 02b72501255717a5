"""The generator's last transposed conv (128 -> 3 channels, 96^2 -> 192^2, batch 32) on the narrow kernel."""
import sys

import torch

sys.path.insert(0, '.')
from sketchyscenecolorization_amd import hip
from sketchyscenecolorization_amd.hip import ACT_RELU, View

x0 = torch.randn(32, 96, 96, 64, device='cuda')
x1 = torch.randn(32, 96, 96, 64, device='cuda')
ab = torch.randn(128, device='cuda')
f = torch.randn(4, 4, 3, 128, device='cuda') * 0.05
out = torch.zeros(32, 192, 192, 8, device='cuda')
v = View(x0, x1, ab, ACT_RELU, None)
fn = lambda: hip.deconv_forward(v, f, out, coff=3, epi=1)
for _ in range(3):
    fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    fn()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 50
print('narrow transposed 128->3: %.1f us  (%.1f MB in: %.1f us at 4 TB/s)' % (ms * 1e3, 2 * x0.numel() * 4 / 1e6, 2 * x0.numel() * 4 / 4e6))
