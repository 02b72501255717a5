"""Which activation / gradient buffers differ between the inline trainer and the overlapped (eager, no graphs) trainer after an
iteration?  (diagnostic for tests/test_gpu_pix2pix.py::test_full_size_overlapped_trainer_equals_inline_trainer_bitwise)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from sketchyscenecolorization_amd.synthetic import synthetic_batch
from sketchyscenecolorization_amd.trainer import GanTrainer
N = 32
a = GanTrainer(img=192, seed=3, max_iter_step=1000, use_graphs=False, overlap_real=False)
b = GanTrainer(img=192, seed=3, max_iter_step=1000, use_graphs=False, real_ahead=False)
ds = [synthetic_batch(N, 100 + k, 192) for k in range(3)]
gs = [synthetic_batch(N, 200 + k, 192) for k in range(3)]
for it in range(3):
    bd, bg = ds[it % 3], gs[it % 3]
    a.d_step(bd, it)
    b.d_step(bd, it)
    torch.cuda.synchronize()
    bad = []
    for k, t in a.bufs._b.items():
        u = b.bufs._b.get(k)
        if u is not None and not torch.equal(t, u):
            d = (t.float() - u.float()).abs()
            bad.append((k[0], float(d.max()), int((d > 0).sum()), t.numel()))
    gbad = [(n, float((a.store.grad(n) - b.store.grad(n)).abs().max())) for n in a.store.names()
            if n.startswith('discriminator') and not n.endswith('/u') and not torch.equal(a.store.grad(n), b.store.grad(n))]
    print('it %d after D-step: differing buffers %d, differing D gradients %s' % (it, len(bad), gbad))
    for x in sorted(bad)[:40]:
        print('    %-28s max %.3e  differing elements %d / %d' % x)
    a.g_step(bg, it)
    b.g_step(bg, it)
    torch.cuda.synchronize()
    wbad = [n for n in a.store.names() if not torch.equal(a.store[n], b.store[n])]
    print('it %d after G-step: differing variables %d' % (it, len(wbad)))
