import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pix2pix as O
from tests.test_gpu_pix2pix import make, relerr
n, img = int(sys.argv[1]), int(sys.argv[2])
p, tr, b, dev = make(n, img)
r = O.build_single_graph_f64(p, **b)
ld = tr.d_step(dev, 0)
print('loss_d', float(ld), float(r['loss_d']))
for k, g in r['grad_d'].items():
    print('%-60s %.3e  ref|max| %.3e' % (k, relerr(tr.store.discriminator.g[k], g), float(g.abs().max())))
tr.store.load_dict(p)
lg = tr.g_step(dev, 0)
print('loss_g', float(lg), float(r['loss_g']))
for k, g in r['grad_g'].items():
    print('%-60s %.3e  ref|max| %.3e' % (k, relerr(tr.store.generator.g[k], g), float(g.abs().max())))
