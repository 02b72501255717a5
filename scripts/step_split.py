"""How the replayed train iteration divides into its two graphs: D-step (incl. the run-ahead generator forward of the G-step that
follows) and G-step, each timed with HIP events over 40 iterations after warm-up."""
import os
import sys
os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sketchyscenecolorization_amd.synthetic import synthetic_batch
from sketchyscenecolorization_amd.trainer import Pix2PixTrainer
tr = Pix2PixTrainer(img=192, seed=0)
bd, bg = synthetic_batch(32, 1, 192), synthetic_batch(32, 2, 192)
for i in range(30):
    tr.train_iteration(bd, bg, i)
torch.cuda.synchronize()
n = 40
ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(n)]
for i in range(n):
    ev[i][0].record()
    tr.d_step(bd, 30 + i, ahead=bg)
    ev[i][1].record()
    tr.g_step(bg, 30 + i, use_ahead=True)
    ev[i][2].record()
torch.cuda.synchronize()
d = sorted(e[0].elapsed_time(e[1]) for e in ev)[n // 2]
g = sorted(e[1].elapsed_time(e[2]) for e in ev)[n // 2]
print('D-step %.3f ms   G-step %.3f ms   sum %.3f ms (medians)' % (d, g, d + g))
