"""Freeze small end-to-end outputs of the CPU oracle (oracle/pix2pix.py) so that drift of the oracle
itself is caught and the HIP path can be checked against stored numbers: seeds regenerate the inputs
and weights, the outputs are stored.  (The reference ships no golden vectors -- parity unpinned.)"""
import os

import numpy as np
import torch

import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import pix2pix as O  # noqa: E402

torch.set_num_threads(4)
img, n = 64, 2
p = O.init_params(0, img=img)
b = O.synthetic_batch(n, seed=42, img=img)
gen = O.generate_pix2pix(p, b['sketches'], b['text'], b['noise_vec'])
r = O.build_single_graph_f64(p, **b)
sel = ['generator/encoder_1/conv/filter', 'generator/encoder_4/scale', 'generator/TextLSTM/embedding',
       'generator/decoder_3/deconv/filter', 'generator/decoder_1/deconv/filter',
       'discriminator/layer_1/conv/filter', 'discriminator/layer_4/offset', 'discriminator/fully_connected/weights']
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'pix2pix_img64_n2_seed0_42.npz'),
                    gen=gen.numpy().astype(np.float32), loss_g=np.float64(r['loss_g']), loss_d=np.float64(r['loss_d']),
                    real_logit=r['real_logit'].numpy(), fake_disc=r['fake_disc'].numpy(),
                    **{'l2_' + k.replace('/', '.'): np.float64(r['grad_g' if k.startswith('gen') else 'grad_d'][k].norm())
                       for k in sel},
                    **{'sum_' + k.replace('/', '.'): np.float64(r['grad_g' if k.startswith('gen') else 'grad_d'][k].sum())
                       for k in sel})
print('ok', float(r['loss_g']), float(r['loss_d']))
