"""Every implicit-GEMM launch of one step of a workload with its GEMM shape, event-timed eagerly (bench.py's PROFILE hook):
usage: shape_table.py bg768 | bg768_train | residual | mru | pix2pix   -> rows 'kernel M N K us TFLOP/s', slowest classes first."""
import os
import sys
from collections import defaultdict

os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sketchyscenecolorization_amd import hip

wl = sys.argv[1] if len(sys.argv) > 1 else 'bg768'
torch.manual_seed(0)
if wl == 'bg768':
    from sketchyscenecolorization_amd.params import Buffers, ParamStore
    from sketchyscenecolorization_amd.residual import ResidualGenerator
    store = ParamStore('BG', 18, 768, 'cuda', 0)
    gen = ResidualGenerator(store, Buffers('cuda'), 'bg')
    x = torch.rand(4, 768, 768, 3, device='cuda') * 2 - 1
    text = torch.randint(1, 18, (4, 8), dtype=torch.int32).numpy()
    step = lambda: gen.forward(x, text, None, 'bg')
elif wl == 'bg768_train':
    from sketchyscenecolorization_amd.bg_colorization import BGTrainer
    tr = BGTrainer(image_size=768)
    tr.use_graphs = False
    x = torch.rand(1, 768, 768, 3, device='cuda') * 2 - 1
    y = torch.rand(1, 768, 768, 3, device='cuda') * 2 - 1
    text = torch.randint(1, 18, (1, 8), dtype=torch.int32).numpy()
    lab = torch.randint(0, 3, (1, 768, 768), dtype=torch.int32, device='cuda')
    step = lambda: tr.train_step(x, y, text, lab)
else:
    from sketchyscenecolorization_amd.synthetic import synthetic_batch
    from sketchyscenecolorization_amd.trainer import GanTrainer
    bt = {'residual': 'Residual', 'mru': 'MRU', 'pix2pix': 'Pix2Pix'}[wl]
    tr = GanTrainer(img=192, seed=0, block_type=bt)
    bd, bg = synthetic_batch(32, 1, 192), synthetic_batch(32, 2, 192)
    step = lambda: tr.train_iteration(bd, bg, counter=0)
for _ in range(2):
    step()
torch.cuda.synchronize()
prof = []
hip.PROFILE = prof
step()
torch.cuda.synchronize()
hip.PROFILE = None
agg = defaultdict(lambda: [0, 0.0, 0.0])
for name, fl, e0, e1, shape, nb in prof:
    a = agg[(name,) + tuple(shape)]
    a[0] += 1
    a[1] += e0.elapsed_time(e1) * 1e3
    a[2] += fl
tot = sum(a[1] for a in agg.values())
print('# %s: %d launches, %.2f ms of event-timed implicit GEMM' % (wl, len(prof), tot / 1e3))
for k, (n, us, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    print('%-26s M=%-8d N=%-5d K=%-6d x%-3d %8.1f us each %7.1f TF/s  %5.1f %% of the time' % (
        k[0], k[1], k[2], k[3], n, us / n, fl / us / 1e6, 100 * us / tot))
