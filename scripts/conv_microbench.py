"""Single-layer microbenchmark of the implicit-GEMM kernel (for rocprofv3 --pmc passes).
usage: conv_microbench.py <layer> [iters] [batch];  layers: enc2 enc3 enc4 d4 dec3 dg3 wg3 dec1 dl1g logit logitd logitw d1 d1p e1 dec1g"""
import sys
import torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sketchyscenecolorization_amd import hip
from sketchyscenecolorization_amd.hip import View
layer = sys.argv[1] if len(sys.argv) > 1 else 'enc2'
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
N = int(sys.argv[3]) if len(sys.argv) > 3 else 32
g = torch.Generator(device='cuda').manual_seed(0)
r = lambda *s: torch.randn(*s, device='cuda', generator=g)
if layer in ('enc2', 'enc3', 'enc4', 'd4'):
    hin, ci, co, stride = {'enc2': (96, 64, 128, 2), 'enc3': (48, 128, 256, 2), 'enc4': (24, 256, 512, 2),
                           'd4': (24, 256, 512, 1)}[layer]
    x, w = r(N, hin, hin, ci), r(4, 4, ci, co) * 0.02
    ab = torch.cat([torch.ones(ci, device='cuda'), torch.zeros(ci, device='cuda')])
    ho = hin // 2 if stride == 2 else hin - 1
    out = torch.empty(N, ho, ho, co, device='cuda')
    fn = lambda: hip.conv_forward(View(x, None, ab, 2), w, stride, 1, out)
    flops = 2.0 * N * ho * ho * co * 16 * ci
elif layer == 'dec3':
    x0, x1, f = r(N, 24, 24, 256), r(N, 24, 24, 256), r(4, 4, 128, 512) * 0.02
    out = torch.empty(N, 48, 48, 128, device='cuda')
    fn = lambda: hip.deconv_forward(View(x0, x1, None, 1, None), f, out)
    flops = 2.0 * N * 48 * 48 * 128 * 4 * 512
elif layer == 'dg3':      # data gradient of encoder_3 (4 sub-pixel phases of 2x2 taps, NK filter, no norm / activation on load)
    dy, w = r(N, 24, 24, 256), r(4, 4, 128, 256) * 0.02
    dx = torch.empty(N, 48, 48, 128, device='cuda')
    fn = lambda: hip.conv_dgrad(View(dy), w, 2, 1, dx)
    flops = 2.0 * N * 48 * 48 * 128 * 4 * 256
elif layer in ('wg3', 'wg3p', 'wg3n', 'wg3m'):   # weight gradient of encoder_3; p: plain operands, n: norm + lrelu on x, m: + mask on dy
    x, dy, dw = r(N, 48, 48, 128), r(N, 24, 24, 256), torch.empty(4, 4, 128, 256, device='cuda')
    ab = torch.cat([torch.ones(128, device='cuda'), torch.zeros(128, device='cuda')])
    ab2 = torch.cat([torch.ones(256, device='cuda'), torch.zeros(256, device='cuda')])
    xv = {'wg3': View(x, None, None, 2), 'wg3p': View(x), 'wg3n': View(x, None, ab, 2), 'wg3m': View(x, None, ab, 2)}[layer]
    gv = View(dy) if layer != 'wg3m' else View(dy, None, ab2, 2)
    fn = lambda: hip.conv_wgrad(xv, gv, dw, 2, 1)
    flops = 2.0 * N * 24 * 24 * 256 * 16 * 128
elif layer == 'dec1':     # generator's last transposed conv 2 x 64 -> 3 (narrow kernel, folded norm + relu on both sources)
    x0, x1, f = r(N, 96, 96, 64), r(N, 96, 96, 64), r(4, 4, 3, 128) * 0.02
    ab = torch.cat([torch.ones(64, device='cuda'), torch.zeros(64, device='cuda')])
    out = torch.empty(N, 192, 192, 4, device='cuda')
    fn = lambda: hip.deconv_forward(View(x0, x1, ab, 1, ab), f, out, nstore=4, epi=1)
    flops = 2.0 * N * 192 * 192 * 3 * 4 * 128
elif layer == 'dl1g':     # data gradient of the discriminator's first conv w.r.t. the 3 generated channels
    dy, w = r(N, 96, 96, 64), r(4, 4, 8, 64) * 0.02
    dx = torch.empty(N, 192, 192, 4, device='cuda')
    fn = lambda: hip.conv_dgrad(View(dy), w, 2, 1, dx, n_off=3, nn=3, nstore=4)
    flops = 2.0 * N * 192 * 192 * 3 * 4 * 64
elif layer == 'logit':    # PatchGAN logit conv 512 -> 1, 4x4 stride 1
    x, w = r(N, 23, 23, 512), r(4, 4, 512, 1) * 0.02
    ab = torch.cat([torch.ones(512, device='cuda'), torch.zeros(512, device='cuda')])
    out = torch.empty(N, 22, 22, 4, device='cuda')
    fn = lambda: hip.conv_forward(View(x, None, ab, 2), w, 1, 1, out, nstore=4)
    flops = 2.0 * N * 22 * 22 * 16 * 512
elif layer == 'logitd':   # its data gradient: one-channel dy -> 512 channels
    dy, w = r(N, 22, 22, 4), r(4, 4, 512, 1) * 0.02
    g4 = torch.empty(N, 23, 23, 512, device='cuda')
    fn = lambda: hip.conv_dgrad(View(dy), w, 1, 1, g4, k_real=1)
    flops = 2.0 * N * 22 * 22 * 16 * 512
elif layer == 'logitw':   # its filter gradient
    x, dy, dw = r(N, 23, 23, 512), r(N, 22, 22, 4), torch.empty(4, 4, 512, 1, device='cuda')
    ab = torch.cat([torch.ones(512, device='cuda'), torch.zeros(512, device='cuda')])
    fn = lambda: hip.conv_wgrad(View(x, None, ab, 2), View(dy), dw, 1, 1)
    flops = 2.0 * N * 22 * 22 * 16 * 512
elif layer in ('d1', 'd1p'):    # discriminator's first conv: 8-channel input (6 real); d1p: filter padded to 8 rows per tap (row-tap form)
    cr = 6 if layer == 'd1' else 8
    x, w = r(N, 192, 192, 8), r(4, 4, cr, 64) * 0.02
    out = torch.empty(N, 96, 96, 64, device='cuda')
    fn = lambda: hip.conv_forward(View(x), w, 2, 1, out)
    flops = 2.0 * N * 96 * 96 * 64 * 16 * cr
elif layer == 'e1':             # generator's first conv: 4-channel input (3 real)
    x, w = r(N, 192, 192, 4), r(4, 4, 3, 64) * 0.02
    out = torch.empty(N, 96, 96, 64, device='cuda')
    fn = lambda: hip.conv_forward(View(x), w, 2, 1, out)
    flops = 2.0 * N * 96 * 96 * 64 * 16 * 3
elif layer == 'dec1g':          # data gradient of the generator's last transposed conv w.r.t. its first 64 input channels
    dy, f = r(N, 192, 192, 4), r(4, 4, 3, 128) * 0.02
    g0 = torch.empty(N, 96, 96, 64, device='cuda')
    fn = lambda: hip.deconv_dgrad(View(dy), f, g0, n_off=0, nn=64)
    flops = 2.0 * N * 96 * 96 * 64 * 16 * 3
for _ in range(3):
    fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    fn()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
print('%s: %.3f ms  %.1f TFLOP/s' % (layer, ms, flops / ms / 1e9))
if os.environ.get('SSC_MICROBENCH_SUM') == '1':        # a checksum of the output (compare two builds / switches on the same inputs)
    res = [t for t in (locals().get('out'), locals().get('dx'), locals().get('dw'), locals().get('g4'), locals().get('g0')) if t is not None][0]
    print('checksum %s: sum %.6e  abs %.6e' % (layer, float(res.double().sum()), float(res.double().abs().sum())))
