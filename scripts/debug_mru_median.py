"""Median / worst relative L2 of the MRU tower's gradients vs float64 on tie-free (uniform noise) inputs."""
import sys
import numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import test_gpu_mru as T
from oracle import mru as M
for img in (64, 192):
    p, tr, b, dev = T._make_trainer(2, img)
    b['sketches'] = torch.rand(b['sketches'].shape, generator=torch.Generator().manual_seed(1)) * 2 - 1
    dev['sketches'] = b['sketches'].cuda()
    r = M.build_single_graph_f64(p, **b)
    r32 = M.build_single_graph(p, **b) if hasattr(M, 'build_single_graph') else None
    tr.d_step(dev, counter=0)
    ed = T._grad_errors(lambda k: tr.store.discriminator.g[k], r['grad_d'])
    tr.store.load_dict(p)
    tr.g_step(dev, counter=0)
    eg = T._grad_errors(lambda k: tr.store.generator.g[k], r['grad_g'])
    for nm, e in (('D', ed), ('G', eg)):
        w = max(e.items(), key=lambda kv: kv[1])
        print('img', img, nm, 'median %.2e  p90 %.2e  worst %.2e (%s)' % (np.median(list(e.values())), np.percentile(list(e.values()), 90), w[1], w[0]))
    if r32 is not None:
        for nm, key in (('D', 'grad_d'), ('G', 'grad_g')):
            c = T._grad_errors(lambda k: r32[key][k], r[key])
            print('   cpu fp32 oracle vs f64', nm, 'median %.2e worst %.2e' % (np.median(list(c.values())), max(c.values())))
