"""Is the overlapped eager trainer reproducible against ITSELF?  Two overlapped trainers (b, c) and the inline one (a), same seeds:
D gradients after each D-step, b vs c and b vs a."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from sketchyscenecolorization_amd.synthetic import synthetic_batch
from sketchyscenecolorization_amd.trainer import GanTrainer
N = 32
a = GanTrainer(img=192, seed=3, max_iter_step=1000, use_graphs=False, overlap_real=False)
b = GanTrainer(img=192, seed=3, max_iter_step=1000, use_graphs=False, real_ahead=False)
c = GanTrainer(img=192, seed=3, max_iter_step=1000, use_graphs=False, real_ahead=False)
ds = [synthetic_batch(N, 100 + k, 192) for k in range(3)]
gs = [synthetic_batch(N, 200 + k, 192) for k in range(3)]
names = [n for n in a.store.names() if n.startswith('discriminator') and not n.endswith('/u')]
for it in range(3):
    bd, bg = ds[it % 3], gs[it % 3]
    for t in (a, b, c):
        t.d_step(bd, it)
        torch.cuda.synchronize()
    bc = [n for n in names if not torch.equal(b.store.grad(n), c.store.grad(n))]
    ba = [n for n in names if not torch.equal(b.store.grad(n), a.store.grad(n))]
    # the fake pass's own gradients (second buffer) of the two overlapped trainers
    g2 = [n for n, (o, k, shp) in b.store.discriminator.offsets.items()
          if not torch.equal(b.store.discriminator.grad2[o:o + k], c.store.discriminator.grad2[o:o + k])]
    print('it %d  overlapped vs overlapped: %s   overlapped vs inline: %s   fake-pass buffers b vs c: %s' % (it, bc, ba, g2))
    for t in (a, b, c):
        t.g_step(bg, it)
        torch.cuda.synchronize()
    if it == 1:
        for n in bc:
            x, y = b.store.grad(n), c.store.grad(n)
            d = (x - y).abs().reshape(-1, x.shape[-1])
            nz = (d > 0).nonzero()
            rows, cols = nz[:, 0], nz[:, 1]
            print('   ', n, tuple(x.shape), 'differing', nz.shape[0], 'of', d.numel(), 'max', float(d.max()),
                  'rows %d..%d cols %d..%d' % (int(rows.min()), int(rows.max()), int(cols.min()), int(cols.max())),
                  'row tiles', sorted(set((rows // 128).tolist()))[:20], 'col tiles', sorted(set((cols // 128).tolist())))
