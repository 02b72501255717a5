"""Model-construction API of obj_lib/models_collection.py (:896-932 aliases and set_param).

The reference functions add nodes to the default tf.Graph and create variables under
``scope_name``; these run the network eagerly on the MI355X.  Variables live in a per-scope
registry keyed by TF variable names (``get_store``); ``reuse=True`` re-uses them exactly like
tf.variable_scope(reuse=True).  The generator returns ``(image [N,3,H,W], noise_vec [N,256])`` and the
discriminator ``(patch_logits [N,1,h,w], class_logits [N,25])`` as the reference does.
"""
import numpy as np
import torch

from .. import hip
from ..pix2pix import Pix2PixDiscriminator, Pix2PixGenerator
from .config import Config

SIZE = 64
NUM_BLOCKS = 1

model_data_format = None
normalizer_fn_e = normalizer_fn_g = normalizer_fn_d = None
normalizer_params_e = normalizer_params_g = normalizer_params_d = None

_REGISTRY = {}


def reset_default_graph():
    """tf.reset_default_graph(): forget every variable and activation buffer."""
    _REGISTRY.clear()


def get_trainer(block_type='Pix2Pix', vocab_size=58, img=192, seed=0, **kw):
    """The tower (variables + activation buffers + optimizer slots) of the 'default graph',
    created on first use like tf.get_variable."""
    if block_type not in ('Pix2Pix', 'Residual', 'MRU'):
        raise NotImplementedError('block_type %r' % block_type)
    key = (block_type, vocab_size, img)
    if key not in _REGISTRY:
        from ..trainer import GanTrainer
        _REGISTRY[key] = GanTrainer(img=img, vocab_size=vocab_size, seed=seed, sn=Config.sn, block_type=block_type,
                                    **kw)
    return _REGISTRY[key]


def get_store(block_type='Pix2Pix', vocab_size=58, img=192, seed=0):
    t = get_trainer(block_type, vocab_size, img, seed)
    return t.store, t.bufs


def set_param(data_format='NCHW'):
    """models_collection.py:902-911: select the (batch-statistics) normaliser for encoder/generator."""
    global model_data_format, normalizer_fn_e, normalizer_fn_g, normalizer_fn_d
    global normalizer_params_e, normalizer_params_g, normalizer_params_d
    if data_format != 'NCHW':
        raise Exception('unsupported')
    model_data_format = data_format
    normalizer_fn_e = normalizer_fn_g = 'batchnorm'
    normalizer_params_e = {'data_format': data_format}
    normalizer_params_g = {'data_format': data_format}
    normalizer_fn_d = normalizer_params_d = None


def _as_device(x, dtype=torch.float32):
    t = torch.as_tensor(np.asarray(x) if not isinstance(x, torch.Tensor) else x)
    return t.to(device='cuda', dtype=dtype).contiguous()


def generate_pix2pix(z, text_vocab_indices, LSTM_hybrid, output_channel, num_classes, vocab_size, reuse=False,
                     data_format='NCHW', labels=None, scope_name=None, noise_vec=None):
    """models_collection.py:444-538.  ``noise_vec`` may be injected (the reference samples
    tf.random_normal inside the graph, :493, and returns it); it is returned either way."""
    assert data_format == 'NCHW' and output_channel == 3
    z = _as_device(z)
    n, _, h, w = z.shape
    tower = get_trainer('Pix2Pix', vocab_size, h)
    if noise_vec is None:
        noise_vec = torch.randn(n, 256, device='cuda')
    noise_vec = _as_device(noise_vec)
    text = text_vocab_indices.cpu().numpy() if isinstance(text_vocab_indices, torch.Tensor) else np.asarray(text_vocab_indices)
    assert text.shape[0] == n
    tower.G.lstm_hybrid = bool(LSTM_hybrid)
    return tower.generate(z, text, noise_vec), noise_vec


def _tower_vocab(block_type, img, vocab_size=None):
    """The discriminator has no vocabulary of its own, but it lives in the tower its generator registered (keyed by
    vocab_size): the caller's value if given, else the one tower of this block type and size that exists, else Config's."""
    if vocab_size is not None:
        return vocab_size
    have = [k[1] for k in _REGISTRY if k[0] == block_type and k[2] == img]
    if len(have) == 1:
        return have[0]
    if len(have) > 1:
        raise ValueError('towers with vocab_size %s exist for %s at %d: pass vocab_size=' % (sorted(have), block_type, img))
    return getattr(Config, 'vocab_size', 58)


def _discriminate(block_type, discrim_inputs, discrim_targets, reuse, data_format, scope_name, vocab_size=None):
    assert data_format == 'NCHW'
    if type(discrim_targets) is list:
        discrim_targets = discrim_targets[-1]
    a, b = _as_device(discrim_inputs), _as_device(discrim_targets)
    n, _, h, w = a.shape
    tr = get_trainer(block_type, _tower_vocab(block_type, h, vocab_size), h)
    xd = tr.bufs.get((scope_name or 'discriminator') + '/api_xd', (n, h, w, 8), zero_on_alloc=True)
    hip.nchw_to_nhwc(a, xd, 0)
    hip.nchw_to_nhwc(b, xd, 3)
    tr.D.sn = bool(Config.sn)
    sn = tr.D.prepare_sn()
    c = tr.D.forward(xd, sn, (scope_name or 'discriminator') + ('/reuse' if reuse else ''))
    disc = c['disc'][..., 0:1].permute(0, 3, 1, 2).contiguous()
    return disc, c['logits'].clone()


def discriminate_pix2pix(discrim_inputs, discrim_targets, num_classes, labels=None, reuse=False,
                         data_format='NCHW', scope_name=None, vocab_size=None):
    """models_collection.py:789-841: PatchGAN logits [N,1,h,w] + spectral-normed class logits [N,25]."""
    return _discriminate('Pix2Pix', discrim_inputs, discrim_targets, reuse, data_format, scope_name, vocab_size)


def discriminate_residual(discrim_inputs, discrim_targets, num_classes, labels=None, reuse=False,
                          data_format='NCHW', scope_name=None, vocab_size=None):
    """models_collection.py:844-893: five stride-2 bottlenecks -> patch logits [N,1,h/32,w/32] + class logits."""
    return _discriminate('Residual', discrim_inputs, discrim_targets, reuse, data_format, scope_name, vocab_size)


def generate_residual(z, text_vocab_indices, LSTM_hybrid, output_channel, num_classes, vocab_size, reuse=False,
                      data_format='NCHW', labels=None, scope_name=None, noise_vec=None):
    """models_collection.py:579-672: ResNet-50-style [3,4,6,3] bottleneck U-Net generator (forward)."""
    assert data_format == 'NCHW' and output_channel == 3
    z = _as_device(z)
    n, _, h, w = z.shape
    tower = get_trainer('Residual', vocab_size, h)
    if noise_vec is None:
        noise_vec = torch.randn(n, 256, device='cuda')
    noise_vec = _as_device(noise_vec)
    text = text_vocab_indices.cpu().numpy() if isinstance(text_vocab_indices, torch.Tensor) else np.asarray(text_vocab_indices)
    assert text.shape[0] == n
    tower.G.lstm_hybrid = bool(LSTM_hybrid)
    return tower.generate(z, text, noise_vec), noise_vec


def generate_mru(z, text_vocab_indices, LSTM_hybrid, output_channel, num_classes, vocab_size, reuse=False,
                 data_format='NCHW', labels=None, scope_name=None, noise_vec=None):
    """models_collection.py:251-377: the MRU generator (forward).  ``labels`` = class ids [N] select the rows of
    the conditional norms (the reference passes them through ``normalizer_params_*['labels']``, :80-82, 270-272)."""
    assert data_format == 'NCHW' and output_channel == 3
    if labels is None:
        raise ValueError('generate_mru needs labels (image_data_class_id): its norms are class-conditional')
    z = _as_device(z)
    n, _, h, w = z.shape
    tower = get_trainer('MRU', vocab_size, h)
    if noise_vec is None:
        noise_vec = torch.randn(n, 256, device='cuda')
    noise_vec = _as_device(noise_vec)
    text = text_vocab_indices.cpu().numpy() if isinstance(text_vocab_indices, torch.Tensor) else np.asarray(text_vocab_indices)
    assert text.shape[0] == n
    tower.G.lstm_hybrid = bool(LSTM_hybrid)
    return tower.generate(z, text, noise_vec, labels=_as_device(labels, torch.int32)), noise_vec


def discriminate_mru(discrim_inputs, discrim_targets, num_classes, labels=None, reuse=False,
                     data_format='NCHW', scope_name=None, vocab_size=None):
    """models_collection.py:676-786: MRU discriminator (spectral norm everywhere, prelu; the sketch is unused)."""
    return _discriminate('MRU', discrim_inputs, discrim_targets, reuse, data_format, scope_name, vocab_size)


generator_mru = generate_mru
discriminator_mru = discriminate_mru
generator_pix2pix = generate_pix2pix
discriminator_pix2pix = discriminate_pix2pix
generator_residual = generate_residual
discriminator_residual = discriminate_residual
