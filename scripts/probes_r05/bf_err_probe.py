"""Where does the bf16-split kernel's distance from float64 come from?  Error statistics (max, rms) of the split and the exact
kernel on one transposed-conv case, by variant: activation, sources, sub-pixel phase."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import tf_ops as T
from sketchyscenecolorization_amd import hip

def rnd(*shape, seed=0, std=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * std
nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous()
nchw = lambda t: t.permute(0, 3, 1, 2).contiguous()

def case(n, h, c0, c1, co, act, tables):
    a, b = rnd(n, c0, h, h, seed=11), (rnd(n, c1, h, h, seed=12) if c1 else None)
    f = rnd(4, 4, co, c0 + c1, seed=13, std=0.05)
    ab0 = torch.cat([1.0 + 0.1 * rnd(c0, seed=14), 0.2 * rnd(c0, seed=15)]) if tables else None
    ab1 = torch.cat([1.0 + 0.1 * rnd(c1, seed=16), 0.2 * rnd(c1, seed=17)]) if (tables and c1) else None
    def tr(x, ab, c):
        x = x.double()
        if ab is not None:
            x = x * ab[:c].double().view(1, -1, 1, 1) + ab[c:].double().view(1, -1, 1, 1)
        return x
    parts = [tr(a, ab0, c0)] + ([tr(b, ab1, c1)] if c1 else [])
    xin = torch.cat(parts, 1)
    xin = torch.relu(xin) if act == 1 else (T.lrelu(xin, 0.2) if act == 2 else xin)
    ref = T.conv2d_transpose_same_s2(xin, f.double())
    ag, bg, fg = nhwc(a).cuda(), (nhwc(b).cuda() if c1 else None), f.cuda()
    res = {}
    for mode in ('bf16x6', 'fp32'):
        hip.ARITH_BF16 = mode == 'bf16x6'
        out = torch.full((n, 2 * h, 2 * h, co), float('nan'), device='cuda')
        hip.deconv_forward(hip.View(ag, bg, ab0.cuda() if tables else None, act, ab1.cuda() if ab1 is not None else None), fg, out)
        e = (nchw(out).cpu().double() - ref)
        res[mode] = (float(e.abs().max()), float(e.pow(2).mean().sqrt()),
                     [float(e[:, :, py::2, px::2].abs().max()) for py in (0, 1) for px in (0, 1)])
    hip.ARITH_BF16 = True
    print('n=%d h=%d c0=%d c1=%d co=%d act=%d tables=%d scale %.2f' % (n, h, c0, c1, co, act, tables, float(ref.abs().max())))
    for m, (mx, rms, ph) in res.items():
        print('   %-7s max %.2e rms %.2e  per phase max %s' % (m, mx, rms, ' '.join('%.2e' % v for v in ph)))

case(4, 12, 64, 64, 64, 1, 1)
case(4, 12, 64, 64, 64, 0, 1)
case(4, 12, 64, 64, 64, 1, 0)
case(4, 12, 128, 0, 64, 1, 1)
case(4, 12, 64, 64, 64, 0, 0)
case(4, 12, 128, 0, 64, 0, 0)
