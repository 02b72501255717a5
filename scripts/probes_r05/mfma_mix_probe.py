"""Which kernels are perturbed when fp32-MFMA and bf16-MFMA kernels share the chip on two streams?  victim x neighbour matrix;
a victim's result is compared bit for bit with its quiet-chip result."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from sketchyscenecolorization_amd import hip
from sketchyscenecolorization_amd.hip import View
g = torch.Generator(device='cuda').manual_seed(0)
r = lambda *s: torch.randn(*s, device='cuda', generator=g)
N = 32
x3, dy3 = r(N, 48, 48, 128), r(N, 24, 24, 256)
ab3 = torch.cat([1 + 0.1 * r(128), 0.1 * r(128)])
w3 = r(4, 4, 128, 256) * 0.02
dyd3, gd3 = r(N, 24, 24, 256), torch.empty(N, 48, 48, 128, device='cuda')
out_c = torch.empty(N, 24, 24, 256, device='cuda')
def set_arith(bf):
    hip.ARITH_BF16 = bf
kern = {
    'wgrad128 (fp32 MFMA)': (lambda out: hip.conv_wgrad(View(x3, None, ab3, 2), View(dy3), out, 2, 1), (4, 4, 128, 256), None),
    'conv fwd fp32 MFMA': (lambda out: hip.conv_forward(View(x3, None, ab3, 2), w3, 2, 1, out), (N, 24, 24, 256), False),
    'conv fwd bf16x6': (lambda out: hip.conv_forward(View(x3, None, ab3, 2), w3, 2, 1, out), (N, 24, 24, 256), True),
    'dgrad fp32 MFMA': (lambda out: hip.conv_dgrad(View(dyd3), w3, 2, 1, out), (N, 48, 48, 128), False),
    'dgrad bf16x6': (lambda out: hip.conv_dgrad(View(dyd3), w3, 2, 1, out), (N, 48, 48, 128), True),
}
xn0, xn1, fn3 = r(N, 96, 96, 64), r(N, 96, 96, 64), r(4, 4, 3, 128) * 0.02
abn = torch.cat([1 + 0.1 * r(64), 0.1 * r(64)])
kern['narrow (packed VALU)'] = (lambda out: hip.deconv_forward(View(xn0, xn1, abn, 1, abn), fn3, out, nstore=4, epi=1), (N, 192, 192, 4), None)
side = torch.cuda.Stream()
scratch = {k: torch.empty(v[1], device='cuda') for k, v in kern.items()}
def run(name, out):
    fn, shape, bf = kern[name]
    if bf is not None:
        set_arith(bf)
    fn(out)
    set_arith(True)
for k in kern:
    run(k, scratch[k])
torch.cuda.synchronize()
for vname, (vfn, vshape, vbf) in kern.items():
    ref = torch.empty(vshape, device='cuda')
    run(vname, ref); torch.cuda.synchronize()
    row = []
    for nname in kern:
        bad = 0
        for rep in range(15):
            out = torch.full(vshape, float('nan'), device='cuda')
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for q in range(4):
                    run(nname, scratch[nname])
            run(vname, out)
            torch.cuda.synchronize()
            bad += int(not torch.equal(out, ref))
        row.append('%s: %d/15' % (nname.split(' (')[0], bad))
    print('victim %-22s beside  %s' % (vname, ' | '.join(row)))
