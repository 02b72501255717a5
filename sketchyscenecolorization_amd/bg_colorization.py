"""Background_Colorization generator (bg_colorization_main.py:302-420) behind the reference's function name.

Only ``create_residual_generator`` (forward) is built: BASELINE.json config 5, the 768x768 large-activation stress
case.  The BG module's discriminator, losses, feed_dict trainer and PNG writer are out of scope (SURVEY.md 8, row
A13 "next").  The reference builds the variables under tf.variable_scope('generator') (:582-586); they live in a
``ParamStore('BG')`` keyed by those TF names so a converted checkpoint loads with ``store.load_dict``.
"""
import numpy as np
import torch

from . import hip
from .params import Buffers, ParamStore
from .residual import ResidualGenerator

_TOWERS = {}


def get_tower(image_size=768, vocab_size=18, ngf=64, seg_classes=3, seed=0, device='cuda'):
    key = (image_size, vocab_size, ngf, seg_classes)
    if key not in _TOWERS:
        if ngf != 64:
            raise NotImplementedError('ngf=%d: the parameter registry is laid out for the default ngf=64' % ngf)
        hip.lib()
        store = ParamStore('BG', vocab_size, image_size, device, seed)
        bufs = Buffers(device)
        _TOWERS[key] = (store, bufs, ResidualGenerator(store, bufs, 'bg', True, ngf, seg_classes))
    return _TOWERS[key]


def reset():
    _TOWERS.clear()


def create_residual_generator(generator_inputs, generator_outputs_channels, vocab_indices, ngf=64, vocab_size=18,
                              seg_classes=3, multi_residual=True):
    """generator_inputs NHWC [N,H,W,3] in [-1,1], vocab_indices int [N,T] ->
    (outputs NHWC [N,H,W,3] = tanh image, region_mask_logits NHWC [N,H,W,seg_classes])."""
    if not multi_residual or generator_outputs_channels != 3:
        raise NotImplementedError('only multi_residual=True, 3 output channels (the reference defaults, :805-807)')
    x = torch.as_tensor(np.asarray(generator_inputs) if not isinstance(generator_inputs, torch.Tensor)
                        else generator_inputs).to(device='cuda', dtype=torch.float32).contiguous()
    text = vocab_indices.cpu().numpy() if isinstance(vocab_indices, torch.Tensor) else np.asarray(vocab_indices)
    store, bufs, gen = get_tower(x.shape[1], vocab_size, ngf, seg_classes)
    ctx = gen.forward(x, text, None, 'bg')
    return ctx['image'], ctx['region_logits']
