"""Which part of the bf16 form moves the MRU discriminator's scalar prelu-leak gradients?  One float64 oracle evaluation, then the
D-step with the bf16 form switched off selectively (forward KN launches / NK launches = data gradients / filter gradients).
usage: python scripts/probes_r05/mru_prelu_grad_probe.py [img]"""
import os, sys
os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import mru as M, pix2pix as O
from sketchyscenecolorization_amd import hip
from sketchyscenecolorization_amd.trainer import GanTrainer
img, n = int(sys.argv[1]) if len(sys.argv) > 1 else 192, 2
p = M.init_params(0, with_discriminator=True, img=img)
b = O.synthetic_batch(n, seed=987 + n, img=img)
b['sketches'] = torch.rand(b['sketches'].shape, generator=torch.Generator().manual_seed(1)) * 2 - 1
dev = {k: (v.cuda() if k != 'text' else v.numpy()) for k, v in b.items()}
r = M.build_single_graph_f64(p, **b)
r32 = M.build_single_graph(p, **b)
big = max(float(g.norm()) for g in r['grad_d'].values())
print('largest |g| %.3e' % big)
attach0, wgrad_exact = hip._attach_split, [False]
mode = ['all']
def attach(d, w):
    if mode[0] == 'no_kn' and d.bmode == 0: return
    if mode[0] == 'no_nk' and d.bmode == 1: return
    if mode[0] == 'none': return
    return attach0(d, w)
hip._attach_split = attach
conv_wgrad0 = hip.conv_wgrad
def rel(a, g):
    a, g = a.detach().cpu().double(), g.detach().cpu().double()
    return float((a - g).norm() / max(float(g.norm()), 1e-4 * big))
for m, wg in [('all', True), ('no_kn', True), ('no_nk', True), ('none', True), ('none', False), ('all', False)]:
    mode[0] = m
    tr = GanTrainer(img=img, seed=1, block_type='MRU')
    tr.use_graphs = False
    tr.store.load_dict(p)
    if not wg:          # exact filter gradients: ARITH_BF16 is read when the descriptor is filled
        def cw(*a, **k):
            keep = hip.ARITH_BF16
            hip.ARITH_BF16 = False
            try:
                return conv_wgrad0(*a, **k)
            finally:
                hip.ARITH_BF16 = keep
        hip.conv_wgrad = cw
    else:
        hip.conv_wgrad = conv_wgrad0
    ld = tr.d_step(dev, counter=0)
    errs = sorted(((rel(tr.store.discriminator.g[k].reshape(g.shape), g), rel(r32['grad_d'][k], g), k) for k, g in r['grad_d'].items()
                   if float(g.norm()) > 1e-12), reverse=True)
    print('--- bf16 convs: %s, bf16 filter gradients: %s; loss_d %.9f (f64 %.9f)' % (m, wg, float(ld), float(r['loss_d'])))
    for e in errs[:4]:
        print('   %.3e (cpu32 %.3e) %s' % e, flush=True)
