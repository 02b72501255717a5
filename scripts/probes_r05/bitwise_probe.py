"""Inline trainer vs overlapped / graph-replayed trainer (tests/test_gpu_pix2pix.py::test_full_size_overlapped_trainer_equals_
inline_trainer_bitwise) with diagnostics: which variables differ, by how much, from which iteration on."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from sketchyscenecolorization_amd.synthetic import synthetic_batch
from sketchyscenecolorization_amd.trainer import GanTrainer
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
graphs = (sys.argv[2] != '0') if len(sys.argv) > 2 else True
a = GanTrainer(img=192, seed=3, max_iter_step=1000, use_graphs=False, overlap_real=False)
b = GanTrainer(img=192, seed=3, max_iter_step=1000, use_graphs=graphs, real_ahead=True)
ds = [synthetic_batch(N, 100 + k, 192) for k in range(3)]
gs = [synthetic_batch(N, 200 + k, 192) for k in range(3)]
for it in range(6):
    bd, bg = ds[it % 3], gs[it % 3]
    la = (float(a.d_step(bd, it)), float(a.g_step(bg, it)))
    lg, ld = b.train_iteration(bd, bg, it, next_batch_d=ds[(it + 1) % 3])
    torch.cuda.synchronize()
    bad = [(n, float((a.store[n] - b.store[n]).abs().max())) for n in a.store.names() if not torch.equal(a.store[n], b.store[n])]
    print('versions', b.store.generator.flat._version, b.store.discriminator.flat._version, 'splits', len(__import__('sketchyscenecolorization_amd').hip._SPLITS))
    print('it %d  loss d %.9g / %.9g  g %.9g / %.9g  differing variables %d %s' % (it, la[0], float(ld), la[1], float(lg), len(bad), bad[:4]))
    if it == 1: print('   differing:', [n.replace('generator/', 'G/').replace('discriminator/', 'D/') for n, _ in bad]); print('   equal:', [n.replace('generator/', 'G/').replace('discriminator/', 'D/') for n in a.store.names() if torch.equal(a.store[n], b.store[n])])
