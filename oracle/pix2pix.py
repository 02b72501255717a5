"""Pix2Pix-variant generator / discriminator / losses restated on torch-CPU fp32.

Oracle, test infrastructure only (see oracle/__init__.py; parity unpinned).
Follows /root/reference/Foreground_Instance_Colorization/obj_lib:
  models_collection.py:408-441  image_encoder_pix2pix
  models_collection.py:444-538  generate_pix2pix
  models_collection.py:150-248  encode_feat_with_text
  models_collection.py:789-841  discriminate_pix2pix
  mru.py:52-92                  fully_connected
  graph_single.py:317-581       get_losses (live branch: sn=True, proj_d=False)
  graph_single.py:221-314       build_single_graph (D-step / G-step gradients)

Parameters live in a flat ``dict`` keyed by the TF variable names the reference
graph would create, in TF layouts.
"""
import math
from collections import OrderedDict

import numpy as np
import torch

from . import tf_ops as T

SIZE = 64           # models_collection.py:10
NUM_CLASSES = 25    # input_pipeline.py:11
T_STEPS = 15        # main_procedure.py:503


# ---------------------------------------------------------------------------
# parameter construction (reference initialisers)
# ---------------------------------------------------------------------------
def generator_param_shapes(vocab_size=58, size=SIZE):
    s = OrderedDict()
    enc = [(3, size), (size, size * 2), (size * 2, size * 4), (size * 4, size * 8), (size * 8, size * 8)]
    for k, (ci, co) in enumerate(enc, start=1):
        s['generator/encoder_%d/conv/filter' % k] = (4, 4, ci, co)
        if k > 1:
            s['generator/encoder_%d/offset' % k] = (co,)
            s['generator/encoder_%d/scale' % k] = (co,)
    c = size * 8
    s['generator/TextLSTM/embedding'] = (vocab_size, c)
    s['generator/TextLSTM/RNN/WLSTM/multi_rnn_cell/cell_0/basic_lstm_cell/kernel'] = (2 * c, 4 * c)
    s['generator/TextLSTM/RNN/WLSTM/multi_rnn_cell/cell_0/basic_lstm_cell/bias'] = (4 * c,)
    s['generator/TextLSTM/RNN/ALSTM/multi_rnn_cell/cell_0/basic_lstm_cell/kernel'] = (4 * c, 4 * c)
    s['generator/TextLSTM/RNN/ALSTM/multi_rnn_cell/cell_0/basic_lstm_cell/bias'] = (4 * c,)
    return s, c


def init_params(seed=0, vocab_size=58, img=192, size=SIZE, num_classes=NUM_CLASSES):
    """All generator + discriminator variables with the reference initialisers."""
    g = torch.Generator().manual_seed(seed)

    def normal(shape, mean, std):
        return torch.randn(shape, generator=g) * std + mean

    def uniform(shape, lo, hi):
        return torch.rand(shape, generator=g) * (hi - lo) + lo

    def glorot(shape):
        lim = math.sqrt(6.0 / (shape[0] + shape[1]))
        return uniform(shape, -lim, lim)

    p = OrderedDict()
    shapes, c = generator_param_shapes(vocab_size, size)
    for name, shp in shapes.items():
        if name.endswith('filter'):
            p[name] = normal(shp, 0.0, 0.02)
        elif name.endswith('offset'):
            p[name] = torch.zeros(shp)
        elif name.endswith('scale'):
            p[name] = normal(shp, 1.0, 0.02)
        elif name.endswith('embedding'):
            p[name] = uniform(shp, -0.08, 0.08)
        elif name.endswith('kernel'):
            p[name] = glorot(shp)
        elif name.endswith('bias'):
            p[name] = torch.zeros(shp)
    hw = img // 32
    cd = c // 8
    p['generator/fully_connected/weights'] = glorot((256, cd * hw * hw))
    p['generator/fully_connected/biases'] = torch.zeros(cd * hw * hw)
    dec = [(5, c + cd, size * 8), (4, size * 16, size * 4), (3, size * 8, size * 2), (2, size * 4, size),
           (1, size * 2, 3)]
    for k, ci, co in dec:
        p['generator/decoder_%d/deconv/filter' % k] = normal((4, 4, co, ci), 0.0, 0.02)
        if k > 1:
            p['generator/decoder_%d/offset' % k] = torch.zeros(co)
            p['generator/decoder_%d/scale' % k] = normal((co,), 1.0, 0.02)
    # discriminator (models_collection.py:789-841)
    dl = [(1, 6, size), (2, size, size * 2), (3, size * 2, size * 4), (4, size * 4, size * 8), (5, size * 8, 1)]
    for k, ci, co in dl:
        p['discriminator/layer_%d/conv/filter' % k] = normal((4, 4, ci, co), 0.0, 0.02)
        if 2 <= k <= 4:
            p['discriminator/layer_%d/offset' % k] = torch.zeros(co)
            p['discriminator/layer_%d/scale' % k] = normal((co,), 1.0, 0.02)
    p['discriminator/fully_connected/weights'] = glorot((size * 8, num_classes))
    p['discriminator/fully_connected/biases'] = torch.zeros(num_classes)
    # sn.py:18 truncated normal, non-trainable
    u = torch.randn((1, num_classes), generator=g)
    while bool((u.abs() > 2).any()):
        r = torch.randn((1, num_classes), generator=g)
        u = torch.where(u.abs() > 2, r, u)
    p['discriminator/fully_connected/u'] = u
    return p


def trainable(params, scope):
    return [k for k in params if k.startswith(scope + '/') and not k.endswith('/u')]


# ---------------------------------------------------------------------------
# generator
# ---------------------------------------------------------------------------
def image_encoder_pix2pix(p, x):
    outs = [T.conv2d_valid_pad(x, p['generator/encoder_1/conv/filter'], 2, 1)]
    for k in range(2, 6):
        r = T.lrelu(outs[-1], 0.2)
        cv = T.conv2d_valid_pad(r, p['generator/encoder_%d/conv/filter' % k], 2, 1)
        outs.append(T.batchnorm(cv, p['generator/encoder_%d/scale' % k], p['generator/encoder_%d/offset' % k]))
    return outs


def encode_feat_with_text(p, visual_encoded, vocab_indices, scope='generator/TextLSTM'):
    """models_collection.py:150-248, per-sample loop, op order as written (the BG module's copy,
    bg_colorization_main.py:117-214, is the same graph under scope 'mLSTM_G')."""
    n, c, vh, vw = visual_encoded.shape
    emb = p[scope + '/embedding']
    kw = p[scope + '/RNN/WLSTM/multi_rnn_cell/cell_0/basic_lstm_cell/kernel']
    bw = p[scope + '/RNN/WLSTM/multi_rnn_cell/cell_0/basic_lstm_cell/bias']
    ka = p[scope + '/RNN/ALSTM/multi_rnn_cell/cell_0/basic_lstm_cell/kernel']
    ba = p[scope + '/RNN/ALSTM/multi_rnn_cell/cell_0/basic_lstm_cell/bias']
    outs = []
    for i in range(n):
        vis = visual_encoded[i:i + 1].permute(0, 2, 3, 1)           # [1,h,w,C]
        vis = T.l2_normalize(vis, 3)
        state_w = torch.zeros(1, 2 * c)
        state_a = torch.zeros(vh * vw, 2 * c)
        h_a = torch.zeros(vh * vw, c)
        for t in range(vocab_indices.shape[1]):
            tok = int(vocab_indices[i, t])
            if tok == 0:        # tf.cond(... == 0, f1, f2): pad tokens skip both LSTMs (:235)
                continue
            w_emb = emb[tok].reshape(1, c)
            h_w, state_w = T.basic_lstm_cell(w_emb, state_w, kw, bw)
            lang = T.l2_normalize(h_w.reshape(1, 1, 1, c), 3).expand(1, vh, vw, c)
            w_feat = w_emb.reshape(1, 1, 1, c).expand(1, vh, vw, c)
            feat_all = torch.cat([vis, w_feat, lang], dim=3).reshape(vh * vw, 3 * c)
            h_a, state_a = T.basic_lstm_cell(feat_all, state_a, ka, ba)
        o = h_a.reshape(1, vh, vw, c)
        o = (torch.log(1.0 + 1e-3 + o) - torch.log(1.0 + 1e-3 - o)) * 0.5
        o = torch.relu(o).permute(0, 3, 1, 2)
        outs.append(o)
    return torch.cat(outs, dim=0)


def fully_connected(x, w, b, activation=None):
    y = x @ w + b
    return activation(y) if activation is not None else y


def generate_pix2pix(p, z, text_vocab_indices, noise_vec, lstm_hybrid=True, return_all=False):
    """generate_pix2pix with ``noise_vec`` injected (the reference samples it
    inside the graph, models_collection.py:493, and returns it)."""
    enc = image_encoder_pix2pix(p, z)
    e5 = enc[-1]
    n, c, hh, ww = e5.shape
    feat = encode_feat_with_text(p, e5, text_vocab_indices) if lstm_hybrid else e5
    cd = c // 8
    noise = fully_connected(noise_vec, p['generator/fully_connected/weights'],
                            p['generator/fully_connected/biases'], T.miu_relu)
    noise = noise.reshape(n, cd, hh, ww)
    layers = list(enc)
    x = torch.cat([feat, noise], dim=1)
    for k in (5, 4, 3, 2):
        if k != 5:
            x = torch.cat([layers[-1], enc[k - 1]], dim=1)
        y = T.conv2d_transpose_same_s2(torch.relu(x), p['generator/decoder_%d/deconv/filter' % k])
        y = T.batchnorm(y, p['generator/decoder_%d/scale' % k], p['generator/decoder_%d/offset' % k])
        layers.append(y)
    x = torch.cat([layers[-1], enc[0]], dim=1)
    out = torch.tanh(T.conv2d_transpose_same_s2(torch.relu(x), p['generator/decoder_1/deconv/filter']))
    if return_all:
        return out, {'enc': enc, 'feat': feat, 'noise': noise, 'dec': layers[5:]}
    return out


# ---------------------------------------------------------------------------
# discriminator
# ---------------------------------------------------------------------------
def discriminate_pix2pix(p, discrim_inputs, discrim_targets, sn=True, return_u=False):
    x = torch.cat([discrim_inputs, discrim_targets], dim=1)
    h = T.lrelu(T.conv2d_valid_pad(x, p['discriminator/layer_1/conv/filter'], 2, 1), 0.2)
    for k, stride in ((2, 2), (3, 2), (4, 1)):
        cv = T.conv2d_valid_pad(h, p['discriminator/layer_%d/conv/filter' % k], stride, 1)
        h = T.lrelu(T.batchnorm(cv, p['discriminator/layer_%d/scale' % k], p['discriminator/layer_%d/offset' % k]), 0.2)
    disc = T.conv2d_valid_pad(h, p['discriminator/layer_5/conv/filter'], 1, 1)
    img = h.mean(dim=(2, 3))
    w = p['discriminator/fully_connected/weights']
    u_new = None
    if sn:
        w, u_new = T.spectral_normed_weight(w, p['discriminator/fully_connected/u'])
    logits = img @ w + p['discriminator/fully_connected/biases']
    if return_u:
        return disc, logits, u_new
    return disc, logits


# ---------------------------------------------------------------------------
# losses (graph_single.py:317-581, Config.sn=True, Config.proj_d=False)
# ---------------------------------------------------------------------------
def regularization_loss(p, scope):
    """ly.l2_regularizer(1e-6) on fully_connected/weights only (mru.py:55,60)."""
    w = p[scope + '/fully_connected/weights']
    return 1e-6 * (w * w).sum() / 2.0


def get_losses(p, images, image_gens, class_id, class_id_d, real_disc, fake_disc, real_logit, fake_logit,
               reg=None):
    reg = reg or regularization_loss
    loss_g_gan = T.softplus(-fake_disc).mean()
    loss_d_gan = T.softplus(fake_disc).mean() + T.softplus(-real_disc).mean()
    ce_real = T.sparse_softmax_ce(real_logit, class_id_d)
    p_true = torch.softmax(real_logit, dim=1).gather(1, class_id_d.long().reshape(-1, 1)).reshape(-1)
    loss_ac_d = ((1.0 - p_true) ** 2.0 * ce_real).mean()
    loss_ac_g = 0.5 * T.sparse_softmax_ce(fake_logit, class_id).mean()
    a = (images - image_gens).abs()
    smooth = torch.where(a < 1.0, 0.5 * a ** 2, a - 0.5).mean()
    loss_g = loss_g_gan + loss_ac_g + 100.0 * smooth + reg(p, 'generator')
    loss_d = loss_d_gan + loss_ac_d + reg(p, 'discriminator')
    parts = {'GAN_loss_g': loss_g_gan, 'GAN_loss_d': loss_d_gan, 'ACGAN_loss_g': loss_ac_g,
             'ACGAN_loss_d': loss_ac_d, 'l1_perceptual_loss': smooth}
    return loss_g, loss_d, parts


def build_single_graph(p, images, sketches, images_d, class_id, class_id_d, text, noise_vec, training=True,
                       lstm_hybrid=True, generator=None, discriminator=None):
    """One tower: forward, both losses and both gradient sets (autograd).  ``generator`` / ``discriminator``
    default to the Pix2Pix pair; oracle.residual passes its own (graph_single.py:244-262 picks them by block_type)."""
    generator = generator or generate_pix2pix
    discriminator = discriminator or discriminate_pix2pix
    if not training:
        with torch.no_grad():
            return generator(p, sketches, text, noise_vec, lstm_hybrid), images, sketches
    q = OrderedDict((k, v.detach().clone().requires_grad_(not k.endswith('/u'))) for k, v in p.items())
    gen = generator(q, sketches, text, noise_vec, lstm_hybrid)
    real_disc, real_logit, u_new = discriminator(q, sketches, images_d, return_u=True)
    fake_disc, fake_logit = discriminator(q, sketches, gen)
    loss_g, loss_d, parts = get_losses(q, images, gen, class_id, class_id_d, real_disc, fake_disc,
                                       real_logit, fake_logit)
    g_names = trainable(q, 'generator')
    d_names = trainable(q, 'discriminator')
    grads_g = torch.autograd.grad(loss_g, [q[k] for k in g_names], retain_graph=True, allow_unused=True)
    grads_d = torch.autograd.grad(loss_d, [q[k] for k in d_names], allow_unused=True)
    grad_g = OrderedDict((k, (g if g is not None else torch.zeros_like(q[k])).detach())
                         for k, g in zip(g_names, grads_g))
    grad_d = OrderedDict((k, (g if g is not None else torch.zeros_like(q[k])).detach())
                         for k, g in zip(d_names, grads_d))
    return {'loss_g': loss_g.detach(), 'loss_d': loss_d.detach(), 'grad_g': grad_g, 'grad_d': grad_d,
            'gen': gen.detach(), 'u_new': u_new.detach(), 'parts': {k: v.detach() for k, v in parts.items()},
            'real_disc': real_disc.detach(), 'fake_disc': fake_disc.detach(),
            'real_logit': real_logit.detach(), 'fake_logit': fake_logit.detach()}


def build_single_graph_f64(p, generator=None, discriminator=None, **batch):
    """The same restatement evaluated in float64: the arbiter when two fp32 paths disagree
    (batch-stat-norm gradients are ill-conditioned: fp32 torch differs from fp64 by up to ~7e-3)."""
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        p64 = OrderedDict((k, v.double()) for k, v in p.items())
        b64 = {k: (v.double() if torch.is_tensor(v) and v.dtype == torch.float32 else v) for k, v in batch.items()}
        return build_single_graph(p64, generator=generator, discriminator=discriminator, **b64)
    finally:
        torch.set_default_dtype(old)


class TrainState(object):
    """Adam slots + step counts for the two optimizers (graph_single.py:138-142)."""

    def __init__(self, params):
        self.v = OrderedDict((k, torch.zeros_like(v)) for k, v in params.items() if not k.endswith('/u'))
        self.t_g = 0
        self.t_d = 0


def d_step(p, st, batch, lr_d, counter, max_iter, **nets):
    """sess.run([opt_d, loss_d]) (main_procedure.py:202-216)."""
    r = build_single_graph(p, **batch, **nets)
    st.t_d += 1
    lr = lr_d * T.lr_decay(counter, max_iter)
    for k, g in r['grad_d'].items():
        T.tf_adam_update(p[k], g, st.v[k], st.t_d, lr)
    return r


def g_step(p, st, batch, lr_g, counter, max_iter, **nets):
    """sess.run([opt_g, loss_g, ...]); SN ``u`` assigns run with opt_g
    (graph_single.py:178-210).  Forward uses the pre-update ``u``."""
    r = build_single_graph(p, **batch, **nets)
    st.t_g += 1
    lr = lr_g * T.lr_decay(counter, max_iter)
    for k, g in r['grad_g'].items():
        T.tf_adam_update(p[k], g, st.v[k], st.t_g, lr)
    p['discriminator/fully_connected/u'] = r['u_new'].clone()
    return r


# ---------------------------------------------------------------------------
# synthetic inputs -- deterministic, numpy RNG
# ---------------------------------------------------------------------------
def synthetic_batch(n, seed=1234, img=192, vocab_size=58, num_classes=NUM_CLASSES, t_steps=T_STEPS):
    rng = np.random.RandomState(seed)
    sk = np.ones((n, 1, img, img), dtype=np.float32)
    for i in range(n):
        for _ in range(6):                      # random 2px poly-lines, ~5% of pixels
            y, x = rng.randint(8, img - 8, size=2)
            for _ in range(img):
                sk[i, 0, y:y + 2, x:x + 2] = -1.0
                y = int(np.clip(y + rng.randint(-2, 3), 0, img - 2))
                x = int(np.clip(x + rng.randint(-2, 3), 0, img - 2))
    sketches = np.repeat(sk, 3, axis=1)
    images = rng.uniform(-1, 1, size=(n, 3, img, img)).astype(np.float32)
    images_d = rng.uniform(-1, 1, size=(n, 3, img, img)).astype(np.float32)
    class_id = rng.randint(0, num_classes, size=n).astype(np.int32)
    class_id_d = rng.randint(0, num_classes, size=n).astype(np.int32)
    text = np.zeros((n, t_steps), dtype=np.int32)
    for i in range(n):
        ln = rng.randint(4, 11)
        text[i, t_steps - ln:] = rng.randint(2, vocab_size, size=ln)
    noise = rng.randn(n, 256).astype(np.float32)
    return {'images': torch.from_numpy(images), 'sketches': torch.from_numpy(sketches),
            'images_d': torch.from_numpy(images_d), 'class_id': torch.from_numpy(class_id),
            'class_id_d': torch.from_numpy(class_id_d), 'text': torch.from_numpy(text),
            'noise_vec': torch.from_numpy(noise)}
