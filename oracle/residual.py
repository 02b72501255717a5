"""Residual-bottleneck generators restated on torch-CPU (oracle; TEST INFRASTRUCTURE ONLY -- nothing under
``sketchyscenecolorization_amd/`` may import this).

PARITY UNPINNED: TensorFlow is absent from this image and the reference ships no golden tensors, so this
restatement is checked only against its own frozen fixtures (tests/golden) and TF's documented op semantics.

Two graphs share the bottleneck blocks:
  * FG ``generate_residual``            models_collection.py:541-672 + residual_util.py:16-171   (NCHW at the API)
  * BG ``create_residual_generator``    bg_colorization_main.py:302-420, 41-98, 217-299          (NHWC at the API)
Tensors are NCHW inside this file; filters keep their TF layouts.
"""
import math
from collections import OrderedDict

import torch

from . import tf_ops as T
from .pix2pix import encode_feat_with_text, fully_connected

UNITS = [3, 4, 6, 3]        # models_collection.py:607, bg_colorization_main.py:315


# ---------------------------------------------------------------------------
# variable shapes in graph-creation order
# ---------------------------------------------------------------------------
def _bn(s, prefix, c):
    s[prefix + '/offset'] = (c,)
    s[prefix + '/scale'] = (c,)


def _en_shapes(s, pre, cin, cout):          # residual_util.py:81-109
    c4 = int(round(cout / 4))
    s[pre + '/block_1/conv/filter'] = (4, 4, cin, c4); _bn(s, pre + '/block_1/batchnorm', c4)
    s[pre + '/block_2/conv_ex/filter'] = (3, 3, c4, c4); _bn(s, pre + '/block_2/batchnorm', c4)
    s[pre + '/block_3/conv_ex/filter'] = (1, 1, c4, cout); _bn(s, pre + '/block_3/batchnorm', cout)
    s[pre + '/block_add/conv/filter'] = (4, 4, cin, cout); _bn(s, pre + '/block_add/batchnorm', cout)


def _de_shapes(s, pre, cin, cout):          # residual_util.py:112-146
    c4 = int(round(cout / 4))
    s[pre + '/block_1/deconv/filter'] = (4, 4, c4, cin); _bn(s, pre + '/block_1/batchnorm', c4)
    s[pre + '/block_2/conv_ex/filter'] = (3, 3, c4, c4); _bn(s, pre + '/block_2/batchnorm', c4)
    s[pre + '/block_3/conv_ex/filter'] = (1, 1, c4, cout); _bn(s, pre + '/block_3/batchnorm', cout)
    s[pre + '/block_add/deconv/filter'] = (4, 4, cout, cin); _bn(s, pre + '/block_add/batchnorm', cout)


def _pu_shapes(s, pre, c):                  # residual_util.py:149-171
    c4 = int(round(c / 4))
    s[pre + '/block_1/conv_ex/filter'] = (4, 4, c, c4); _bn(s, pre + '/block_1/batchnorm', c4)
    s[pre + '/block_2/conv_ex/filter'] = (3, 3, c4, c4); _bn(s, pre + '/block_2/batchnorm', c4)
    s[pre + '/block_3/conv_ex/filter'] = (1, 1, c4, c); _bn(s, pre + '/block_3/batchnorm', c)


def _lstm_shapes(s, scope, vocab, c):
    s[scope + '/embedding'] = (vocab, c)
    for cell, rows in (('WLSTM', 2 * c), ('ALSTM', 4 * c)):
        base = scope + '/RNN/%s/multi_rnn_cell/cell_0/basic_lstm_cell/' % cell
        s[base + 'kernel'] = (rows, 4 * c)
        s[base + 'bias'] = (4 * c,)


def generator_shapes(kind, vocab_size=None, img=None, size=64, seg_classes=3):
    """kind='fg': generate_residual (192x192 default); kind='bg': create_residual_generator (768x768 default)."""
    fg = kind == 'fg'
    vocab_size = vocab_size if vocab_size is not None else (58 if fg else 18)
    img = img if img is not None else (192 if fg else 768)
    top = size * 8 if fg else size * 16
    enc_c = [size, size * 2, size * 4, size * 8, top]
    bn1 = (lambda pre: pre) if fg else (lambda pre: pre + '/batchnorm')
    s = OrderedDict()
    s['generator/encoder_1/conv_ex/filter'] = (7, 7, 3, size); _bn(s, bn1('generator/encoder_1'), size)
    for k in range(2, 6):
        _en_shapes(s, 'generator/encoder_%d_0' % k, enc_c[k - 2], enc_c[k - 1])
        for u in range(1, UNITS[k - 2]):
            _pu_shapes(s, 'generator/encoder_%d_%d' % (k, u), enc_c[k - 1])
    if fg:
        _lstm_shapes(s, 'generator/TextLSTM', vocab_size, top)
        hw = img // 32
        s['generator/fully_connected/weights'] = (256, top // 8 * hw * hw)
        s['generator/fully_connected/biases'] = (top // 8 * hw * hw,)
    else:
        _lstm_shapes(s, 'generator/mLSTM_G', vocab_size, top)
        s['generator/region_br_projection/conv_ex/filter'] = (1, 1, top, seg_classes)
        _bn(s, 'generator/region_br_projection/batchnorm', seg_classes)
    dec_out = {5: size * 8, 4: size * 4, 3: size * 2, 2: size}
    cur = top + (top // 8 if fg else 0)
    for k in (5, 4, 3, 2):
        cin = cur if k == 5 else dec_out[k + 1] + enc_c[k - 1]
        _de_shapes(s, 'generator/decoder_%d_0' % k, cin, dec_out[k])
        for u in range(1, UNITS[k - 2]):
            _pu_shapes(s, 'generator/decoder_%d_%d' % (k, u), dec_out[k])
        if not fg:
            s['generator/region_br_%d/deconv/filter' % k] = (4, 4, seg_classes, seg_classes)
            _bn(s, 'generator/region_br_%d/batchnorm' % k, seg_classes)
    s['generator/decoder_1/deconv/filter'] = (4, 4, 3, size * 2); _bn(s, bn1('generator/decoder_1'), 3)
    if not fg:
        s['generator/region_br_1/deconv/filter'] = (4, 4, seg_classes, seg_classes)
        _bn(s, 'generator/region_br_1/batchnorm', seg_classes)
    return s


def discriminator_shapes(size=64, num_classes=25):
    """discriminate_residual (models_collection.py:844-893): 5 stride-2 encoder bottlenecks, a 4x4 s1 SAME patch
    head and the spectral-normed class head on layer_4's mean."""
    s = OrderedDict()
    chans = [(6, size), (size, size * 2), (size * 2, size * 4), (size * 4, size * 8), (size * 8, size * 8)]
    for k, (ci, co) in enumerate(chans, start=1):
        _en_shapes(s, 'discriminator/layer_%d' % k, ci, co)
    s['discriminator/layer_5/conv_ex/filter'] = (4, 4, size * 8, 1)
    s['discriminator/fully_connected/weights'] = (size * 8, num_classes)
    s['discriminator/fully_connected/biases'] = (num_classes,)
    s['discriminator/fully_connected/u'] = (1, num_classes)
    return s


def init_params(kind, seed=0, with_discriminator=False, **kw):
    """Reference initialisers: filters N(0,0.02), scale N(1,0.02), offset 0, embedding U(-0.08,0.08),
    LSTM kernels / FC weights glorot-uniform, biases 0."""
    g = torch.Generator().manual_seed(seed)
    p = OrderedDict()
    shapes = generator_shapes(kind, **kw)
    if with_discriminator:
        shapes.update(discriminator_shapes())
    for name, shp in shapes.items():
        leaf = name.rsplit('/', 1)[1]
        if leaf == 'filter':
            p[name] = torch.randn(shp, generator=g) * 0.02
        elif leaf == 'scale':
            p[name] = torch.randn(shp, generator=g) * 0.02 + 1.0
        elif leaf in ('offset', 'bias', 'biases'):
            p[name] = torch.zeros(shp)
        elif leaf == 'embedding':
            p[name] = torch.rand(shp, generator=g) * 0.16 - 0.08
        elif leaf in ('kernel', 'weights'):
            lim = math.sqrt(6.0 / (shp[0] + shp[1]))
            p[name] = torch.rand(shp, generator=g) * 2 * lim - lim
        elif leaf == 'u':           # sn.py:18 truncated normal, non-trainable
            u = torch.randn(shp, generator=g)
            while bool((u.abs() > 2).any()):
                u = torch.where(u.abs() > 2, torch.randn(shp, generator=g), u)
            p[name] = u
        else:
            raise ValueError(name)
    return p


# ---------------------------------------------------------------------------
# blocks (residual_util.py / bg_colorization_main.py:217-299)
# ---------------------------------------------------------------------------
def _norm(p, pre, x):
    return T.batchnorm(x, p[pre + '/scale'], p[pre + '/offset'])


def bottleneck_residual_en(p, pre, x, stride=2):
    orig = x
    x = T.lrelu(_norm(p, pre + '/block_1/batchnorm', T.conv2d_valid_pad(x, p[pre + '/block_1/conv/filter'], stride, 1)), 0.2)
    x = T.lrelu(_norm(p, pre + '/block_2/batchnorm', T.conv2d_same(x, p[pre + '/block_2/conv_ex/filter'], 1)), 0.2)
    x = _norm(p, pre + '/block_3/batchnorm', T.conv2d_same(x, p[pre + '/block_3/conv_ex/filter'], 1))
    if stride != 1:
        orig = _norm(p, pre + '/block_add/batchnorm', T.conv2d_valid_pad(orig, p[pre + '/block_add/conv/filter'], stride, 1))
    return T.lrelu(x + orig, 0.2)


def bottleneck_residual_de(p, pre, x, need_relu=True):
    orig = x
    x = torch.relu(_norm(p, pre + '/block_1/batchnorm', T.conv2d_transpose_same_s2(x, p[pre + '/block_1/deconv/filter'])))
    x = torch.relu(_norm(p, pre + '/block_2/batchnorm', T.conv2d_same(x, p[pre + '/block_2/conv_ex/filter'], 1)))
    x = _norm(p, pre + '/block_3/batchnorm', T.conv2d_same(x, p[pre + '/block_3/conv_ex/filter'], 1))
    orig = _norm(p, pre + '/block_add/batchnorm', T.conv2d_transpose_same_s2(orig, p[pre + '/block_add/deconv/filter']))
    x = x + orig
    return torch.relu(x) if need_relu else x


def bottleneck_residual_pu(p, pre, x, is_encoder):
    act = (lambda t: T.lrelu(t, 0.2)) if is_encoder else torch.relu
    orig = x
    x = act(_norm(p, pre + '/block_1/batchnorm', T.conv2d_same(x, p[pre + '/block_1/conv_ex/filter'], 1)))
    x = act(_norm(p, pre + '/block_2/batchnorm', T.conv2d_same(x, p[pre + '/block_2/conv_ex/filter'], 1)))
    x = _norm(p, pre + '/block_3/batchnorm', T.conv2d_same(x, p[pre + '/block_3/conv_ex/filter'], 1))
    return act(x + orig)


def _encoder(p, x, bn1):
    """image_encoder_residual (models_collection.py:541-576) == bg_colorization_main.py:317-340."""
    out = T.lrelu(_norm(p, bn1('generator/encoder_1'), T.conv2d_same(x, p['generator/encoder_1/conv_ex/filter'], 2)), 0.2)
    layers = [out]
    for k in range(2, 6):
        out = bottleneck_residual_en(p, 'generator/encoder_%d_0' % k, layers[-1], 2)
        for u in range(1, UNITS[k - 2]):
            out = bottleneck_residual_pu(p, 'generator/encoder_%d_%d' % (k, u), out, True)
        layers.append(out)
    return layers


def discriminate_residual(p, discrim_inputs, discrim_targets, sn=True, return_u=False):
    """models_collection.py:844-893 (Config.sn=True: only the class head's weights are spectral-normed)."""
    x = torch.cat([discrim_inputs, discrim_targets], dim=1)
    layers = []
    for k in range(1, 5):
        x = bottleneck_residual_en(p, 'discriminator/layer_%d' % k, x, 2)
        layers.append(x)
    rectified = x
    convolved = bottleneck_residual_en(p, 'discriminator/layer_5', rectified, 2)
    disc = T.conv2d_same(convolved, p['discriminator/layer_5/conv_ex/filter'], 1)
    img = rectified.mean(dim=(2, 3))
    w = p['discriminator/fully_connected/weights']
    u_new = None
    if sn:
        w, u_new = T.spectral_normed_weight(w, p['discriminator/fully_connected/u'])
    logits = img @ w + p['discriminator/fully_connected/biases']
    if return_u:
        return disc, logits, u_new
    return disc, logits


def build_single_graph(p, **batch):
    """One Residual tower (graph_single.py:221-314 with block_type='Residual'): losses + both gradient sets."""
    from .pix2pix import build_single_graph as bsg
    return bsg(p, generator=generate_residual, discriminator=discriminate_residual, **batch)


def build_single_graph_f64(p, **batch):
    from .pix2pix import build_single_graph_f64 as bsg64
    return bsg64(p, generator=generate_residual, discriminator=discriminate_residual, **batch)


def generate_residual(p, z, text_vocab_indices, noise_vec, lstm_hybrid=True, return_all=False):
    """FG generate_residual (models_collection.py:579-672); ``noise_vec`` injected (sampled in-graph at :626)."""
    bn1 = lambda pre: pre
    layers = _encoder(p, z, bn1)
    e5 = layers[-1]
    n, c, hh, ww = e5.shape
    feat = encode_feat_with_text(p, e5, text_vocab_indices) if lstm_hybrid else e5
    noise = fully_connected(noise_vec, p['generator/fully_connected/weights'], p['generator/fully_connected/biases'],
                            T.miu_relu).reshape(n, c // 8, hh, ww)
    n_enc = len(layers)
    for dl in range(4):
        skip = n_enc - dl - 1
        k = skip + 1
        inp = torch.cat([feat, noise], 1) if dl == 0 else torch.cat([layers[-1], layers[skip]], 1)
        out = bottleneck_residual_de(p, 'generator/decoder_%d_0' % k, inp)
        for u in range(1, UNITS[skip - 1]):
            out = bottleneck_residual_pu(p, 'generator/decoder_%d_%d' % (k, u), out, False)
        layers.append(out)
    inp = torch.cat([layers[-1], layers[0]], 1)
    out = torch.tanh(_norm(p, 'generator/decoder_1', T.conv2d_transpose_same_s2(inp, p['generator/decoder_1/deconv/filter'])))
    if return_all:
        return out, {'layers': layers, 'feat': feat, 'noise': noise}
    return out


def create_residual_generator(p, generator_inputs, vocab_indices, return_all=False):
    """BG create_residual_generator (bg_colorization_main.py:302-420).  generator_inputs NHWC [N,H,W,3];
    returns (image NHWC [N,H,W,3], region logits NHWC [N,H,W,seg])."""
    bn1 = lambda pre: pre + '/batchnorm'
    x = generator_inputs.permute(0, 3, 1, 2)
    layers = _encoder(p, x, bn1)
    feat = encode_feat_with_text(p, layers[-1], vocab_indices, scope='generator/mLSTM_G')
    reg = torch.relu(_norm(p, 'generator/region_br_projection/batchnorm',
                           T.conv2d_same(layers[-1], p['generator/region_br_projection/conv_ex/filter'], 1)))
    n_enc = len(layers)
    for dl in range(4):
        skip = n_enc - dl - 1
        k = skip + 1
        inp = feat if dl == 0 else torch.cat([layers[-1], layers[skip]], 1)
        out = bottleneck_residual_de(p, 'generator/decoder_%d_0' % k, inp)
        for u in range(1, UNITS[skip - 1]):
            out = bottleneck_residual_pu(p, 'generator/decoder_%d_%d' % (k, u), out, False)
        layers.append(out)
        reg = torch.relu(_norm(p, 'generator/region_br_%d/batchnorm' % k,
                               T.conv2d_transpose_same_s2(reg, p['generator/region_br_%d/deconv/filter' % k])))
    inp = torch.cat([layers[-1], layers[0]], 1)
    out = torch.tanh(_norm(p, 'generator/decoder_1/batchnorm',
                           T.conv2d_transpose_same_s2(inp, p['generator/decoder_1/deconv/filter'])))
    reg = torch.relu(_norm(p, 'generator/region_br_1/batchnorm',
                           T.conv2d_transpose_same_s2(reg, p['generator/region_br_1/deconv/filter'])))
    img, seg = out.permute(0, 2, 3, 1).contiguous(), reg.permute(0, 2, 3, 1).contiguous()
    if return_all:
        return img, seg, {'layers': layers, 'feat': feat}
    return img, seg


# ---------------------------------------------------------------------------
# Background_Colorization training graph (bg_colorization_main.py:516-726)
# ---------------------------------------------------------------------------
BG_EPS = 1e-12      # bg_colorization_main.py:21


def bg_discriminator_shapes(ndf=64):
    """create_residual_discriminator (:550-580): five stride-2 encoder bottlenecks, sigmoid on the last."""
    s = OrderedDict()
    chans = [(6, ndf), (ndf, ndf * 2), (ndf * 2, ndf * 4), (ndf * 4, ndf * 8), (ndf * 8, 1024)]
    for k, (ci, co) in enumerate(chans, start=1):
        _en_shapes(s, 'discriminator/layer_%d' % k, ci, co)
    return s


def init_bg_params(seed=0, **kw):
    """Generator + discriminator of the BG module with the reference initialisers."""
    p = init_params('bg', seed=seed, **kw)
    g = torch.Generator().manual_seed(seed + 977)
    for name, shp in bg_discriminator_shapes().items():
        leaf = name.rsplit('/', 1)[1]
        p[name] = (torch.randn(shp, generator=g) * 0.02 if leaf == 'filter' else
                   torch.randn(shp, generator=g) * 0.02 + 1.0 if leaf == 'scale' else torch.zeros(shp))
    return p


def create_residual_discriminator(p, discrim_inputs, discrim_targets):
    """NHWC in, sigmoid probabilities [N, H/32, W/32, 1024] NHWC out."""
    x = torch.cat([discrim_inputs, discrim_targets], dim=3).permute(0, 3, 1, 2)
    for k in range(1, 6):
        x = bottleneck_residual_en(p, 'discriminator/layer_%d' % k, x, 2)
    return torch.sigmoid(x).permute(0, 2, 3, 1)


def bg_losses(outputs, region_logits, predict_real, predict_fake, targets, labels_gt, gan_weight=1.0, l1_weight=100.0,
              seg_weight=100.0):
    """:596-627.  labels_gt int [N,H,W] in {0 = foreground, 1, 2}; the L1 term averages |t - o| over the pixels with
    label != 0 (all three channels)."""
    discrim_loss = (-(torch.log(predict_real + BG_EPS) + torch.log(1 - predict_fake + BG_EPS))).mean()
    gen_loss_gan = (-torch.log(predict_fake + BG_EPS)).mean()
    sel = labels_gt.reshape(-1) != 0
    gen_loss_l1 = (targets - outputs).abs().reshape(-1, outputs.shape[3])[sel].mean()
    seg = T.sparse_softmax_ce(region_logits.reshape(-1, region_logits.shape[3]), labels_gt.reshape(-1)).mean()
    gen_loss = gen_loss_gan * gan_weight + gen_loss_l1 * l1_weight + seg * seg_weight
    return discrim_loss, gen_loss, {'gen_loss_GAN': gen_loss_gan, 'gen_loss_L1': gen_loss_l1, 'region_mask_loss': seg}


def bg_build_graph(p, inputs, targets, text, labels_gt):
    """One evaluation of create_model in train mode: both losses and both gradient sets from ONE forward pass.
    (The reference builds the generator gradients under control_dependencies([discrim_train]); which discriminator
    weights its backward ops then read is not defined by the TF1 ref-variable semantics.  This restatement -- and the
    build -- differentiate both losses at the weights the forward pass used.)"""
    q = OrderedDict((k, v.detach().clone().requires_grad_(True)) for k, v in p.items())
    outputs, region_logits = create_residual_generator(q, inputs, text)
    predict_real = create_residual_discriminator(q, inputs, targets)
    predict_fake = create_residual_discriminator(q, inputs, outputs)
    d_loss, g_loss, parts = bg_losses(outputs, region_logits, predict_real, predict_fake, targets, labels_gt)
    g_names = [k for k in q if k.startswith('generator/')]
    d_names = [k for k in q if k.startswith('discriminator/')]
    gg = torch.autograd.grad(g_loss, [q[k] for k in g_names], retain_graph=True, allow_unused=True)
    gd = torch.autograd.grad(d_loss, [q[k] for k in d_names], allow_unused=True)
    z = lambda k, g: (g if g is not None else torch.zeros_like(q[k])).detach()
    return {'discrim_loss': d_loss.detach(), 'gen_loss': g_loss.detach(),
            'parts': {k: v.detach() for k, v in parts.items()}, 'outputs': outputs.detach(),
            'region_logits': region_logits.detach(), 'predict_real': predict_real.detach(),
            'predict_fake': predict_fake.detach(),
            'grad_g': OrderedDict((k, z(k, g)) for k, g in zip(g_names, gg)),
            'grad_d': OrderedDict((k, z(k, g)) for k, g in zip(d_names, gd))}


def bg_build_graph_f64(p, inputs, targets, text, labels_gt):
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        return bg_build_graph(OrderedDict((k, v.double()) for k, v in p.items()), inputs.double(), targets.double(),
                              text, labels_gt)
    finally:
        torch.set_default_dtype(old)


def bg_learning_rate(lr, step, max_steps):
    """tf.train.polynomial_decay(lr, global_step, decay_steps=round(0.75*max_steps), end=lr/10, power=0.9) (:632-637)."""
    decay_steps = int(round(max_steps * 0.75))
    s = min(step, decay_steps)
    return (lr - lr / 10.0) * (1.0 - s / decay_steps) ** 0.9 + lr / 10.0


class BGTrainState(object):
    def __init__(self, params):
        self.m = OrderedDict((k, torch.zeros_like(v)) for k, v in params.items())
        self.v = OrderedDict((k, torch.zeros_like(v)) for k, v in params.items())
        self.t = 0


def bg_train_step(p, st, batch, step, lr=2e-4, max_steps=100000, beta1=0.5):
    """model.train (:641-655): Adam(lr_t, beta1=0.5, beta2=0.999) on both nets; both optimizers have made the same
    number of steps, so one counter serves."""
    r = bg_build_graph(p, **batch)
    st.t += 1
    lr_t = bg_learning_rate(lr, step, max_steps)
    for k, g in list(r['grad_d'].items()) + list(r['grad_g'].items()):
        T.tf_adam_update(p[k], g, st.v[k], st.t, lr_t, beta1=beta1, beta2=0.999, m=st.m[k])
    return r


def bg_synthetic_batch(n=1, img=96, seed=0, t_steps=8, vocab_size=18):
    g = torch.Generator().manual_seed(seed)
    inputs = torch.rand(n, img, img, 3, generator=g) * 2 - 1
    targets = torch.rand(n, img, img, 3, generator=g) * 2 - 1
    text = torch.zeros(n, t_steps, dtype=torch.int32)
    text[:, t_steps - 5:] = torch.randint(1, vocab_size, (n, 5), generator=g, dtype=torch.int32)
    labels = torch.randint(0, 3, (n, img // 8, img // 8), generator=g).repeat_interleave(8, 1).repeat_interleave(8, 2)
    return {'inputs': inputs, 'targets': targets, 'text': text, 'labels_gt': labels.to(torch.int32)}
