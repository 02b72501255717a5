"""Freeze small outputs of the Residual / BG / MRU oracles (oracle/residual.py, oracle/mru.py): seeds regenerate
inputs and weights, the generated images (64x64 / 96x96) and a few loss / gradient scalars are stored.
(The reference ships no golden vectors -- parity unpinned; these pin the oracle against its own drift and give the
HIP path stored numbers to match.)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import mru as M  # noqa: E402
from oracle import pix2pix as O  # noqa: E402
from oracle import residual as R  # noqa: E402

torch.set_num_threads(4)
HERE = os.path.dirname(os.path.abspath(__file__))


def inputs(img, n=2, seed=42):
    b = O.synthetic_batch(n, seed=seed, img=img)
    return b


def bg_inputs(img=96, n=1, seed=7):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(n, img, img, 3, generator=g) * 2 - 1
    text = torch.zeros(n, 8, dtype=torch.int32)
    text[:, 3:] = torch.randint(1, 18, (n, 5), generator=g, dtype=torch.int32)
    return x, text


def to_f64(p):
    return {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in p.items()}


if __name__ == '__main__':
    b = inputs(64)
    # Residual: forward + one float64 training graph
    p = R.init_params('fg', seed=0, with_discriminator=True, img=64)
    gen = R.generate_residual(p, b['sketches'], b['text'], b['noise_vec'])
    r = R.build_single_graph_f64(p, **b)
    # MRU generator forward (labels = class ids)
    pm = M.init_params(0, img=64)
    gen_m = M.generate_mru(pm, b['sketches'], b['text'], b['class_id'], b['noise_vec'])
    # BG generator forward
    pb = R.init_params('bg', seed=0, img=96)
    x, text = bg_inputs()
    img_b, seg_b = R.create_residual_generator(pb, x, text)
    # the same graph in float64: what a host with other fp32 conv kernels can be held to (the fp32 oracle itself sits up to
    # 2e-3 from it on this 96x96 batch-norm stack, and by how much depends on the CPU's instruction set)
    img_b64, seg_b64 = R.create_residual_generator(to_f64(pb), x.double(), text)
    np.savez_compressed(os.path.join(HERE, 'variants_seed0_42.npz'),
                        residual_gen=gen.numpy(), residual_loss_g=np.float64(r['loss_g']),
                        residual_loss_d=np.float64(r['loss_d']), residual_fake_disc=r['fake_disc'].numpy(),
                        residual_real_logit=r['real_logit'].numpy(), mru_gen=gen_m.numpy(),
                        bg_image=img_b.numpy(), bg_region_logits=seg_b.numpy(),
                        bg_image_f64=img_b64.numpy(), bg_region_logits_f64=seg_b64.numpy())
    print('ok', float(r['loss_g']), float(r['loss_d']))
