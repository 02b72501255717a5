"""When do the branches of the replayed D-step / G-step graphs really run?  One-lane timestamp kernels (hip.mark) captured into
the graphs at the forks, joins and section ends; no profiler attached.  Prints the marks of one replayed iteration in time order."""
import os
import sys
os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sketchyscenecolorization_amd import hip
from sketchyscenecolorization_amd.synthetic import synthetic_batch
from sketchyscenecolorization_amd.trainer import Pix2PixTrainer
hip.MARKS = {}
tr = Pix2PixTrainer(img=192, seed=0, use_graphs=True)
bd, bg = synthetic_batch(32, 1, 192), synthetic_batch(32, 2, 192)
bd, bg = tr.input_buffers('d', bd), tr.input_buffers('g', bg)
for i in range(30):
    tr.train_iteration(bd, bg, i, next_batch_d=bd)
torch.cuda.synchronize()
for rep in range(2):
    hip._mark_buf.zero_()
    torch.cuda.synchronize()
    tr.train_iteration(bd, bg, 40 + rep, next_batch_d=bd)
    torch.cuda.synchronize()
    print('--- replayed iteration %d' % rep)
    for n, t in hip.read_marks().items():
        print('%9.1f us  %s' % (t, n))
