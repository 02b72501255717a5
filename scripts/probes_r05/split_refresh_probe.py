"""After a train iteration every registered filter's bf16 planes (refreshed by the batched launch behind the optimizer) must equal a
fresh single-filter split of the current weights."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from sketchyscenecolorization_amd import hip
from sketchyscenecolorization_amd.synthetic import synthetic_batch
from sketchyscenecolorization_amd.trainer import GanTrainer
b = GanTrainer(img=192, seed=3, max_iter_step=1000, use_graphs=False, overlap_real=False)
bd, bg = synthetic_batch(8, 100, 192), synthetic_batch(8, 200, 192)
for it in range(2):
    b.d_step(bd, it); b.g_step(bg, it)
torch.cuda.synchronize()
bad = 0
for k, e in hip._SPLITS.items():
    old = e.buf.clone()
    hip._split_launch(e)
    torch.cuda.synchronize()
    if not torch.equal(old, e.buf):
        bad += 1
        diff = (old != e.buf).nonzero()
        print('STALE planes', k[1:], 'first differing byte', int(diff[0]), 'of', e.buf.numel(), 'count', diff.numel())
print('entries', len(hip._SPLITS), 'stale', bad)
