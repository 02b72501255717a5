import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pix2pix as O
from tests.test_gpu_pix2pix import make
n, img = int(sys.argv[1]), int(sys.argv[2])
p, tr, b, dev = make(n, img)
r64 = O.build_single_graph_f64(p, **b)
r32 = O.build_single_graph(p, **b)
tr.d_step(dev, 0)
gd = {k: v.clone().cpu() for k, v in tr.store.discriminator.g.items()}
tr.store.load_dict(p)
tr.g_step(dev, 0)
gg = {k: v.clone().cpu() for k, v in tr.store.generator.g.items()}
def stats(a, ref):
    a = a.double(); d = (a - ref)
    return float(d.abs().max() / ref.abs().max()), float(d.norm() / ref.norm()), tuple(int(i) for i in torch.nonzero(d.abs() == d.abs().max())[0])
for name, ref in list(r64['grad_d'].items()) + list(r64['grad_g'].items()):
    ours = gd[name] if name in gd else gg[name]
    o32 = r32['grad_d'][name] if name in r32['grad_d'] else r32['grad_g'][name]
    m1, l1, w1 = stats(ours, ref); m2, l2, w2 = stats(o32, ref)
    print('%-45s hip: max %.2e l2 %.2e at %s | torch32: max %.2e l2 %.2e' % (name[-45:], m1, l1, w1, m2, l2))
