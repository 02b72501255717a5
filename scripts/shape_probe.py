"""One conv shape in a loop: us per launch and TFLOP/s under the current environment (SSC_FWD_CFG=n pins the tile configuration,
SSC_TS_FORCE="whole tiles per CU,slices" the tail split).  usage: shape_probe.py N H W Cin Cout k stride [bn]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sketchyscenecolorization_amd import hip

n, h, w, ci, co, k, stride = (int(v) for v in sys.argv[1:8])
bn = len(sys.argv) > 8
x = torch.randn(n, h, w, ci, device='cuda')
wt = torch.randn(k, k, ci, co, device='cuda') * 0.05
ab = torch.cat([torch.ones(ci), torch.zeros(ci)]).cuda()
out = torch.empty(n, h // stride, w // stride, co, device='cuda')
scale, offset, a2, s2 = torch.ones(co, device='cuda'), torch.zeros(co, device='cuda'), torch.empty(2 * co, device='cuda'), torch.empty(2 * co, device='cuda')
xv = hip.View(x, None, ab, 1)
fn = lambda: hip.conv_forward(xv, wt, stride, 0, out, same=True, bn=(scale, offset, a2, s2) if bn else None)
for _ in range(3):
    fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    fn()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
fl = 2.0 * out.numel() * k * k * ci
print('cfg=%s ts=%s: %.1f us, %.1f TFLOP/s' % (os.environ.get('SSC_FWD_CFG'), os.environ.get('SSC_TS_FORCE'), us, fl / us / 1e6))
