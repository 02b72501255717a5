"""Is the bf16x6 contraction's error BIASED?  (A rounding that truncates instead of rounding to nearest adds coherently in sums that
cancel -- the MRU discriminator's scalar prelu-leak gradient lost a digit with the bf16 form.)  1x1 convs (= matmuls) and a 3x3
conv through the library in both arithmetics vs float64: mean signed error, rms error, their ratio, for zero-mean and for
positive-mean operands.  usage: python scripts/probes_r05/mfma_bias_probe.py"""
import os, sys
os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from sketchyscenecolorization_amd import hip

def run(x, w, mode):
    hip.ARITH_BF16 = mode == 'bf16x6'
    n, h, _, ci = x.shape
    co = w.shape[3]
    out = torch.empty(n, h, h, co, device='cuda')
    hip.conv_forward(hip.View(x), w, 1, 0, out, same=True)
    torch.cuda.synchronize()
    hip.ARITH_BF16 = True
    return out

g = torch.Generator().manual_seed(0)
for name, k, ci, co, shift_x, shift_w in [('zero-mean 1x1 K=2304', 1, 2304, 512, 0.0, 0.0), ('x>0 1x1 K=2304', 1, 2304, 512, 1.0, 0.0),
                                           ('x>0,w>0 1x1 K=2304', 1, 2304, 512, 1.0, 0.05), ('zero-mean 3x3 K=9*256', 3, 256, 512, 0.0, 0.0),
                                           ('x>0 3x3 K=9*256', 3, 256, 512, 1.0, 0.0), ('x>0,w>0 3x3', 3, 256, 512, 1.0, 0.05)]:
    x = (torch.randn(2, 24, 24, ci, generator=g) + shift_x)
    w = (torch.randn(k, k, ci, co, generator=g) * 0.05 + shift_w)
    ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w.permute(3, 2, 0, 1).double(), padding=k // 2).permute(0, 2, 3, 1)
    xg, wg = x.cuda(), w.cuda()
    line = '%-24s scale %.3g' % (name, float(ref.abs().mean()))
    for mode in ('bf16x6', 'fp32'):
        e = run(xg, wg, mode).cpu().double() - ref
        line += ' | %s mean %+.3e rms %.3e ratio %+.3f' % (mode, float(e.mean()), float(e.pow(2).mean().sqrt()), float(e.mean() / e.pow(2).mean().sqrt()))
    print(line, flush=True)
