"""Per-launch table of the implicit-GEMM kernels for one train iteration (shape, ms, TFLOP/s)."""
import sys
import torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sketchyscenecolorization_amd import hip
from sketchyscenecolorization_amd.synthetic import synthetic_batch
from sketchyscenecolorization_amd.trainer import Pix2PixTrainer
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
bt = sys.argv[2] if len(sys.argv) > 2 else 'Pix2Pix'
tr = Pix2PixTrainer(img=192, seed=0, block_type=bt)
bd, bg = synthetic_batch(n, 1, 192), synthetic_batch(n, 2, 192)
for i in range(2):
    tr.train_iteration(bd, bg, i)
torch.cuda.synchronize()
hip.PROFILE = []
ev0 = torch.cuda.Event(enable_timing=True); ev1 = torch.cuda.Event(enable_timing=True)
ev0.record()
tr.d_step(bd, 2)
mark = len(hip.PROFILE)
tr.g_step(bg, 2)
ev1.record()
torch.cuda.synchronize()
prof, hip.PROFILE = hip.PROFILE, None
tot = 0.0
for i, (name, fl, e0, e1, shp, _nb) in enumerate(prof):
    ms = e0.elapsed_time(e1)
    tot += ms
    if i == mark:
        print('---- G-step ----')
    print('%-24s M=%-8d N=%-5d K=%-7d %8.3f ms %7.1f TF  %6.2f GF' % (name, shp[0], shp[1], shp[2], ms, fl / ms / 1e9, fl / 1e9))
print('igemm total %.2f ms of step %.2f ms' % (tot, ev0.elapsed_time(ev1)))
