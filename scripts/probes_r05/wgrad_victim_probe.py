"""Is the 128x128 filter-gradient kernel (wgrad128.hip) perturbed by conv launches running beside it on another stream?
Filter gradient of the discriminator's layer_3 / layer_2 shapes in a loop on one stream, compared bit for bit with a quiet-chip
reference, while another stream runs conv launches (bf16-split or exact fp32: SSC_ARITH)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from sketchyscenecolorization_amd import hip
from sketchyscenecolorization_amd.hip import View
g = torch.Generator(device='cuda').manual_seed(0)
r = lambda *s: torch.randn(*s, device='cuda', generator=g)
N = 32
victims = {}
x3, dy3 = r(N, 48, 48, 128), r(N, 24, 24, 256)
ab3 = torch.cat([1 + 0.1 * r(128), 0.1 * r(128)])
victims['wgrad layer_3'] = (lambda out: hip.conv_wgrad(View(x3, None, ab3, 2), View(dy3), out, 2, 1), (4, 4, 128, 256))
x2, dy2 = r(N, 96, 96, 64), r(N, 48, 48, 128)
victims['wgrad layer_2'] = (lambda out: hip.conv_wgrad(View(x2, None, None, 2), View(dy2), out, 2, 1), (4, 4, 64, 128))
# neighbours: the discriminator's data gradients and forward convs
dyd4, wd4 = r(N, 23, 23, 512), r(4, 4, 256, 512) * 0.02
gd4 = torch.empty(N, 24, 24, 256, device='cuda')
dyd3, wd3 = r(N, 24, 24, 256), r(4, 4, 128, 256) * 0.02
gd3 = torch.empty(N, 48, 48, 128, device='cuda')
dyd2, wd2 = r(N, 48, 48, 128), r(4, 4, 64, 128) * 0.02
gd2 = torch.empty(N, 96, 96, 64, device='cuda')
def neighbours():
    hip.conv_dgrad(View(dyd4), wd4, 1, 1, gd4)
    hip.conv_dgrad(View(dyd3), wd3, 2, 1, gd3)
    hip.conv_dgrad(View(dyd2), wd2, 2, 1, gd2)
side = torch.cuda.Stream()
neighbours(); torch.cuda.synchronize()
for name, (fn, shape) in victims.items():
    ref = torch.empty(shape, device='cuda')
    fn(ref); torch.cuda.synchronize()
    alone = beside = 0
    for rep in range(20):
        out = torch.full(shape, float('nan'), device='cuda')
        fn(out); torch.cuda.synchronize()
        alone += int(not torch.equal(out, ref))
    for rep in range(40):
        out = torch.full(shape, float('nan'), device='cuda')
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            neighbours(); neighbours()
        fn(out)
        torch.cuda.synchronize()
        beside += int(not torch.equal(out, ref))
    print('%-14s differs from the quiet-chip result: alone %d / 20, beside the conv stream %d / 40' % (name, alone, beside))

# how far is the perturbed result from float64, compared with the quiet-chip result?
import torch.nn.functional as F
for name, (fn, shape) in victims.items():
    ref = torch.empty(shape, device='cuda')
    fn(ref); torch.cuda.synchronize()
    outs = []
    for rep in range(4):
        out = torch.full(shape, float('nan'), device='cuda')
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            neighbours(); neighbours()
        fn(out)
        torch.cuda.synchronize()
        outs.append(out.clone())
    if name.endswith('3'):
        xin = F.leaky_relu(x3.double() * ab3[:128].double() + ab3[128:].double(), 0.2)
        dy = dy3.double()
    else:
        xin = F.leaky_relu(x2.double(), 0.2)
        dy = dy2.double()
    xp = F.pad(xin.permute(0, 3, 1, 2), (1, 1, 1, 1))
    cols = F.unfold(xp, 4, stride=2)                      # [N, ci*16, P]
    ci = xin.shape[-1]
    g64 = torch.einsum('nkp,npc->kc', cols, dy.reshape(dy.shape[0], -1, dy.shape[-1]))   # [(ci,kh,kw), co]
    g64 = g64.reshape(ci, 4, 4, -1).permute(1, 2, 0, 3)
    e_q = (ref.double() - g64).abs()
    print('%-14s |grad| max %.3g   quiet-chip error vs f64: max %.3e rms %.3e' % (name, float(g64.abs().max()), float(e_q.max()), float(e_q.pow(2).mean().sqrt())))
    for o in outs:
        e = (o.double() - g64).abs()
        dq = (o - ref).abs()
        print('      beside the conv stream: error vs f64 max %.3e rms %.3e;  vs quiet: max %.3e, differing elements %d / %d, also differs from run 0: %d'
              % (float(e.max()), float(e.pow(2).mean().sqrt()), float(dq.max()), int((dq > 0).sum()), o.numel(), int((o != outs[0]).sum())))
