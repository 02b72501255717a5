import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import mru as M, pix2pix as O, tf_ops as T
from sketchyscenecolorization_amd.trainer import GanTrainer
from sketchyscenecolorization_amd import hip
img, n = 64, 2
p = M.init_params(0, with_discriminator=True, img=img)
tr = GanTrainer(img=img, seed=1, block_type='MRU')
tr.store.load_dict(p)
b = O.synthetic_batch(n, seed=987 + n, img=img)
dev = {k: (v.cuda() if k != 'text' else v.numpy()) for k, v in b.items()}
def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).norm() / max(float(b.norm()), 1e-30))
p64 = {k: v.double() for k, v in p.items()}
gen = b['images'].double().requires_grad_(True)     # any image as the "generated" one
fd, fl = M.discriminate_mru(p64, b['sketches'].double(), gen)
loss = T.softplus(-fd).mean() + 0.5 * T.sparse_softmax_ce(fl, b['class_id']).mean()
g_ref = torch.autograd.grad(loss, gen)[0]
# HIP
N = n
xd = torch.zeros(N, img, img, 8, device='cuda')
hip.nchw_to_nhwc(dev['sketches'], xd, 0); hip.nchw_to_nhwc(dev['images'], xd, 3)
sn = tr.D.prepare_sn()
cf = tr.D.forward(xd, sn, 'df')
print('disc fwd', rel(cf['disc'][..., 0], fd[:, 0]), 'logits', rel(cf['logits'], fl))
loss_g = tr.loss[0:1]; loss_g.zero_()
rows = cf['disc'].shape[0] * cf['disc'].shape[1] * cf['disc'].shape[2]
dl5 = torch.zeros_like(cf['disc'])
hip.call('ssc_softplus_loss', cf['disc'], 4, rows, -1.0, 1.0 / rows, loss_g, dl5, 1.0 / rows)
dlog = torch.zeros(N, 25, device='cuda')
hip.call('ssc_acgan_loss', cf['logits'], dev['class_id'], N, 25, 0, 0.5, loss_g, dlog)
dgen = tr.D.backward(cf, dl5, dlog, sn, False, True, accumulate=False)
torch.cuda.synchronize()
print('loss', float(loss_g), float(loss))
print('dgen rel', rel(dgen[..., :3].permute(0, 3, 1, 2), g_ref), float(g_ref.norm()))
# per-level contributions in the oracle: gradient restricted to pyramid levels
