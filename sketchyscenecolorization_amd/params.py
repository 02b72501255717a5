"""Parameter registry keyed by the TF variable names of the reference graph.

The reference owns its variables in a global tf.Graph under the scopes
``generator`` / ``discriminator`` (graph_single.py:296-299).  Here every scope is
one flat fp32 device buffer (parameters, gradients, Adam second moments) so that
TF-style Adam is a single kernel launch and the data-parallel gradient exchange
is a few large RCCL all-reduces over contiguous memory; named parameters are
views into the flat buffer in their TF layouts (checkpoints map 1:1 to TF names).
"""
import math
from collections import OrderedDict

import numpy as np
import torch

ALIGN = 64  # floats (256 B): keeps every view 16-byte aligned for float4 loads

SIZE = 64           # models_collection.py:10
NUM_CLASSES = 25    # input_pipeline.py:11


def pix2pix_param_specs(vocab_size=58, img=192, num_classes=NUM_CLASSES, size=SIZE):
    """(name, shape, init) for every variable of generate_pix2pix / discriminate_pix2pix
    (models_collection.py:408-538, 789-841), in graph-creation order."""
    g = []
    enc = [(3, size), (size, size * 2), (size * 2, size * 4), (size * 4, size * 8), (size * 8, size * 8)]
    for k, (ci, co) in enumerate(enc, start=1):
        g.append(('generator/encoder_%d/conv/filter' % k, (4, 4, ci, co), ('normal', 0.0, 0.02)))
        if k > 1:
            g.append(('generator/encoder_%d/offset' % k, (co,), ('zeros',)))
            g.append(('generator/encoder_%d/scale' % k, (co,), ('normal', 1.0, 0.02)))
    c = size * 8
    g.append(('generator/TextLSTM/embedding', (vocab_size, c), ('uniform', -0.08, 0.08)))
    for cell, rows in (('WLSTM', 2 * c), ('ALSTM', 4 * c)):
        base = 'generator/TextLSTM/RNN/%s/multi_rnn_cell/cell_0/basic_lstm_cell/' % cell
        g.append((base + 'kernel', (rows, 4 * c), ('glorot',)))
        g.append((base + 'bias', (4 * c,), ('zeros',)))
    hw = img // 32
    cd = c // 8
    g.append(('generator/fully_connected/weights', (256, cd * hw * hw), ('glorot',)))
    g.append(('generator/fully_connected/biases', (cd * hw * hw,), ('zeros',)))
    dec = [(5, c + cd, size * 8), (4, size * 16, size * 4), (3, size * 8, size * 2), (2, size * 4, size),
           (1, size * 2, 3)]
    for k, ci, co in dec:
        g.append(('generator/decoder_%d/deconv/filter' % k, (4, 4, co, ci), ('normal', 0.0, 0.02)))
        if k > 1:
            g.append(('generator/decoder_%d/offset' % k, (co,), ('zeros',)))
            g.append(('generator/decoder_%d/scale' % k, (co,), ('normal', 1.0, 0.02)))
    d = []
    dl = [(1, 6, size), (2, size, size * 2), (3, size * 2, size * 4), (4, size * 4, size * 8), (5, size * 8, 1)]
    for k, ci, co in dl:
        d.append(('discriminator/layer_%d/conv/filter' % k, (4, 4, ci, co), ('normal', 0.0, 0.02)))
        if 2 <= k <= 4:
            d.append(('discriminator/layer_%d/offset' % k, (co,), ('zeros',)))
            d.append(('discriminator/layer_%d/scale' % k, (co,), ('normal', 1.0, 0.02)))
    d.append(('discriminator/fully_connected/weights', (size * 8, num_classes), ('glorot',)))
    d.append(('discriminator/fully_connected/biases', (num_classes,), ('zeros',)))
    nontrainable = [('discriminator/fully_connected/u', (1, num_classes), ('truncated_normal',))]
    return g, d, nontrainable


RESIDUAL_UNITS = (3, 4, 6, 3)       # models_collection.py:607, bg_colorization_main.py:315


def residual_generator_specs(kind='fg', vocab_size=None, img=None, size=SIZE, seg_classes=3):
    """(name, shape, init) of the bottleneck-residual generators in graph-creation order.

    kind='fg': ``generate_residual`` (models_collection.py:541-672, blocks residual_util.py:81-171);
    kind='bg': ``create_residual_generator`` (bg_colorization_main.py:302-420) -- same blocks, 1024-channel
    bottleneck, 'mLSTM_G' caption cells, no noise head, 3-channel region branch, norms under '.../batchnorm'."""
    fg = kind == 'fg'
    vocab_size = vocab_size if vocab_size is not None else (58 if fg else 18)
    img = img if img is not None else (192 if fg else 768)
    top = size * 8 if fg else size * 16
    enc_c = [size, size * 2, size * 4, size * 8, top]
    out = []
    filt, zeros, ones = ('normal', 0.0, 0.02), ('zeros',), ('normal', 1.0, 0.02)

    def bn(pre, c):
        out.append((pre + '/offset', (c,), zeros))
        out.append((pre + '/scale', (c,), ones))

    def unit(pre, first, shape_first, c4, cout, shortcut=None):
        out.append((pre + '/block_1/%s/filter' % first, shape_first, filt)); bn(pre + '/block_1/batchnorm', c4)
        out.append((pre + '/block_2/conv_ex/filter', (3, 3, c4, c4), filt)); bn(pre + '/block_2/batchnorm', c4)
        out.append((pre + '/block_3/conv_ex/filter', (1, 1, c4, cout), filt)); bn(pre + '/block_3/batchnorm', cout)
        if shortcut is not None:
            out.append((pre + '/block_add/%s/filter' % first, shortcut, filt)); bn(pre + '/block_add/batchnorm', cout)

    def lstm(scope, c):
        out.append((scope + '/embedding', (vocab_size, c), ('uniform', -0.08, 0.08)))
        for cell, rows in (('WLSTM', 2 * c), ('ALSTM', 4 * c)):
            base = scope + '/RNN/%s/multi_rnn_cell/cell_0/basic_lstm_cell/' % cell
            out.append((base + 'kernel', (rows, 4 * c), ('glorot',)))
            out.append((base + 'bias', (4 * c,), zeros))

    top_bn = (lambda pre: pre) if fg else (lambda pre: pre + '/batchnorm')
    out.append(('generator/encoder_1/conv_ex/filter', (7, 7, 3, size), filt)); bn(top_bn('generator/encoder_1'), size)
    for k in range(2, 6):
        cin, co = enc_c[k - 2], enc_c[k - 1]
        unit('generator/encoder_%d_0' % k, 'conv', (4, 4, cin, co // 4), co // 4, co, (4, 4, cin, co))
        for u in range(1, RESIDUAL_UNITS[k - 2]):
            unit('generator/encoder_%d_%d' % (k, u), 'conv_ex', (4, 4, co, co // 4), co // 4, co)
    if fg:
        lstm('generator/TextLSTM', top)
        hw = img // 32
        out.append(('generator/fully_connected/weights', (256, top // 8 * hw * hw), ('glorot',)))
        out.append(('generator/fully_connected/biases', (top // 8 * hw * hw,), zeros))
    else:
        lstm('generator/mLSTM_G', top)
        out.append(('generator/region_br_projection/conv_ex/filter', (1, 1, top, seg_classes), filt))
        bn('generator/region_br_projection/batchnorm', seg_classes)
    dec_out = {5: size * 8, 4: size * 4, 3: size * 2, 2: size}
    for k in (5, 4, 3, 2):
        cin = (top + (top // 8 if fg else 0)) if k == 5 else dec_out[k + 1] + enc_c[k - 1]
        co = dec_out[k]
        unit('generator/decoder_%d_0' % k, 'deconv', (4, 4, co // 4, cin), co // 4, co, (4, 4, co, cin))
        for u in range(1, RESIDUAL_UNITS[k - 2]):
            unit('generator/decoder_%d_%d' % (k, u), 'conv_ex', (4, 4, co, co // 4), co // 4, co)
        if not fg:
            out.append(('generator/region_br_%d/deconv/filter' % k, (4, 4, seg_classes, seg_classes), filt))
            bn('generator/region_br_%d/batchnorm' % k, seg_classes)
    out.append(('generator/decoder_1/deconv/filter', (4, 4, 3, size * 2), filt)); bn(top_bn('generator/decoder_1'), 3)
    if not fg:
        out.append(('generator/region_br_1/deconv/filter', (4, 4, seg_classes, seg_classes), filt))
        bn('generator/region_br_1/batchnorm', seg_classes)
    return out


def residual_discriminator_specs(num_classes=NUM_CLASSES, size=SIZE):
    """discriminate_residual (models_collection.py:844-893): (trainable specs, non-trainable specs)."""
    out = []
    filt, zeros, ones = ('normal', 0.0, 0.02), ('zeros',), ('normal', 1.0, 0.02)
    chans = [(6, size), (size, size * 2), (size * 2, size * 4), (size * 4, size * 8), (size * 8, size * 8)]
    for k, (ci, co) in enumerate(chans, start=1):
        pre = 'discriminator/layer_%d' % k
        for blk, shape, c in (('block_1/conv', (4, 4, ci, co // 4), co // 4), ('block_2/conv_ex', (3, 3, co // 4, co // 4), co // 4),
                              ('block_3/conv_ex', (1, 1, co // 4, co), co), ('block_add/conv', (4, 4, ci, co), co)):
            out.append(('%s/%s/filter' % (pre, blk), shape, filt))
            out.append(('%s/%s/batchnorm/offset' % (pre, blk.split('/')[0]), (c,), zeros))
            out.append(('%s/%s/batchnorm/scale' % (pre, blk.split('/')[0]), (c,), ones))
    out.append(('discriminator/layer_5/conv_ex/filter', (4, 4, size * 8, 1), filt))
    out.append(('discriminator/fully_connected/weights', (size * 8, num_classes), ('glorot',)))
    out.append(('discriminator/fully_connected/biases', (num_classes,), zeros))
    return out, [('discriminator/fully_connected/u', (1, num_classes), ('truncated_normal',))]


def bg_discriminator_specs(ndf=SIZE):
    """create_residual_discriminator (bg_colorization_main.py:550-580): five stride-2 encoder bottlenecks."""
    out = []
    filt, zeros, ones = ('normal', 0.0, 0.02), ('zeros',), ('normal', 1.0, 0.02)
    chans = [(6, ndf), (ndf, ndf * 2), (ndf * 2, ndf * 4), (ndf * 4, ndf * 8), (ndf * 8, 1024)]
    for k, (ci, co) in enumerate(chans, start=1):
        pre = 'discriminator/layer_%d' % k
        for blk, shape, c in (('block_1/conv', (4, 4, ci, co // 4), co // 4), ('block_2/conv_ex', (3, 3, co // 4, co // 4), co // 4),
                              ('block_3/conv_ex', (1, 1, co // 4, co), co), ('block_add/conv', (4, 4, ci, co), co)):
            out.append(('%s/%s/filter' % (pre, blk), shape, filt))
            out.append(('%s/%s/batchnorm/offset' % (pre, blk.split('/')[0]), (c,), zeros))
            out.append(('%s/%s/batchnorm/scale' % (pre, blk.split('/')[0]), (c,), ones))
    return out


MRU_ENC_UNITS = [(1, 8, 64), (2, 64, 128), (3, 128, 256), (4, 256, 512)]
MRU_DEC_UNITS = [(0, 512, 384, 67), (2, 384, 256, 131), (4, 256, 128, 67), (6, 128, 128, 11), (8, 128, 64, 3)]


def mru_generator_specs(vocab_size=58, img=192, num_classes=NUM_CLASSES):
    """(name, shape, init) of generate_mru / image_encoder_mru (models_collection.py:68-147, 251-377; blocks
    mru.py:353-461, 527-591) in graph-creation order.  TF uniquifies the default conv scope ('Conv', 'Conv_1', ...);
    biases (TF shape (1,C,1,1)) are stored flat; conditional-norm tables are [num_classes, C]
    (models_collection.py:29-31: offset zeros, scale ones)."""
    out = []
    filt, zeros, ones = ('normal', 0.0, 0.02), ('zeros',), ('ones',)

    def conv(pre, k, cin, cout, norm=False, bias_init=zeros):
        out.append((pre + '/weights', (k, k, cin, cout), filt))
        out.append((pre + '/biases', (cout,), bias_init))
        if norm:
            cbn(pre, cout)

    def cbn(pre, c):
        out.append((pre + '/offset', (num_classes, c), zeros))
        out.append((pre + '/scale', (num_classes, c), ones))

    conv('generator/Conv', 7, 3, 8)
    for u, ch, d in MRU_ENC_UNITS:
        pre = 'generator/mru_conv_unit_t_%d_layer_0' % u
        cbn(pre + '/norm_activation_in', ch)
        conv(pre + '/update_gate', 3, ch + 3, ch, bias_init=('const', 0.5))     # mru.py:360
        conv(pre + '/Conv', 3, 3, ch)
        cbn(pre + '/norm_activation_merge_1', ch)
        conv(pre + '/Conv_1', 3, ch, d, norm=True)
        conv(pre + '/Conv_2', 3, d, d)
        if ch != d:
            conv(pre + '/Conv_3', 1, ch, d)
    cbn('generator/mru_conv_unit_last_norm', 512)
    c = 512
    out.append(('generator/TextLSTM/embedding', (vocab_size, c), ('uniform', -0.08, 0.08)))
    for cell, rows in (('WLSTM', 2 * c), ('ALSTM', 4 * c)):
        base = 'generator/TextLSTM/RNN/%s/multi_rnn_cell/cell_0/basic_lstm_cell/' % cell
        out.append((base + 'kernel', (rows, 4 * c), ('glorot',)))
        out.append((base + 'bias', (4 * c,), zeros))
    hw = img // 16
    out.append(('generator/fully_connected/weights', (256, 64 * hw * hw), ('glorot',)))
    out.append(('generator/fully_connected/biases', (64 * hw * hw,), zeros))
    for u, ch, d, ci in MRU_DEC_UNITS:
        pre = 'generator/mru_deconv_unit_t_%d_layer_0' % u
        conv(pre + '/Conv', 3, ch + ci, ch)
        conv(pre + '/Conv_1', 3, ch + ci, d)
        conv(pre + '/Conv_2', 3, ch + ci, d, norm=True)
        conv(pre + '/Conv_3', 3, d, d, norm=True)
        if ch != d:
            conv(pre + '/Conv_4', 1, ch, d, norm=True)
    conv('generator/Conv_1', 7, 64, 3)
    return out


MRU_DISC_UNITS = [(1, 8, 128), (2, 128, 256), (3, 256, 512), (4, 512, 768)]


def mru_discriminator_specs(num_classes=NUM_CLASSES):
    """discriminate_mru (models_collection.py:676-786) with Config.sn=True: (trainable specs, non-trainable ``u``
    specs).  Every conv / FC weight has its own power-iteration vector (sn.py:17-18); prelu leaks are scalars
    initialised to 0.2 (models_collection.py:56-60)."""
    out, nt = [], []
    filt, zeros = ('normal', 0.0, 0.02), ('zeros',)

    def conv(pre, k, cin, cout, prelu=False, bias_init=zeros):
        out.append((pre + '/weights', (k, k, cin, cout), filt))
        nt.append((pre + '/u', (1, cout), ('truncated_normal',)))
        out.append((pre + '/biases', (cout,), bias_init))
        if prelu:
            out.append((pre + '/prelu/param', (), ('const', 0.2)))

    conv('discriminator/Conv', 7, 3, 8, prelu=True)
    for u, ch, d in MRU_DISC_UNITS:
        pre = 'discriminator/mru_conv_unit_t_%d_layer_0' % u
        out.append((pre + '/norm_activation_in/prelu/param', (), ('const', 0.2)))
        conv(pre + '/update_gate', 3, ch + 3, ch, bias_init=('const', 0.5))
        conv(pre + '/Conv', 3, 3, ch)
        out.append((pre + '/norm_activation_merge_1/prelu/param', (), ('const', 0.2)))
        conv(pre + '/Conv_1', 3, ch, d, prelu=True)
        conv(pre + '/Conv_2', 3, d, d)
        conv(pre + '/Conv_3', 1, ch, d)
    out.append(('discriminator/mru_conv_unit_last_norm/prelu/param', (), ('const', 0.2)))
    conv('discriminator/Conv_1', 1, 768, 1)
    out.append(('discriminator/fully_connected/weights', (768, num_classes), ('glorot',)))
    nt.append(('discriminator/fully_connected/u', (1, num_classes), ('truncated_normal',)))
    out.append(('discriminator/fully_connected/biases', (num_classes,), zeros))
    return out, nt


def _init_tensor(shape, init, gen):
    kind = init[0]
    if kind == 'zeros':
        return torch.zeros(shape)
    if kind == 'ones':
        return torch.ones(shape)
    if kind == 'const':
        return torch.full(shape, float(init[1]))
    if kind == 'normal':
        return torch.randn(shape, generator=gen) * init[2] + init[1]
    if kind == 'uniform':
        return torch.rand(shape, generator=gen) * (init[2] - init[1]) + init[1]
    if kind == 'glorot':        # xavier/glorot uniform (mru.py:54; TF default for BasicLSTMCell kernels)
        lim = math.sqrt(6.0 / (shape[0] + shape[1]))
        return torch.rand(shape, generator=gen) * 2 * lim - lim
    if kind == 'truncated_normal':      # sn.py:18
        t = torch.randn(shape, generator=gen)
        while bool((t.abs() > 2).any()):
            t = torch.where(t.abs() > 2, torch.randn(shape, generator=gen), t)
        return t
    raise ValueError(kind)


class Scope(object):
    """One variable scope = flat parameter / gradient / Adam-v buffers + named views."""

    def __init__(self, name, specs, device):
        self.name = name
        self.specs = specs
        self.offsets = OrderedDict()
        off = 0
        for n, shape, _ in specs:
            self.offsets[n] = (off, int(np.prod(shape)), tuple(shape))
            off += (int(np.prod(shape)) + ALIGN - 1) // ALIGN * ALIGN
        self.numel = off
        self.flat = torch.zeros(off, dtype=torch.float32, device=device)
        self.grad = torch.zeros(off, dtype=torch.float32, device=device)
        self.adam_v = torch.zeros(off, dtype=torch.float32, device=device)
        self.adam_m = None      # first moment: only optimizers with beta1 != 0 (the BG module) allocate it
        self.adam_t = 0
        self.p = OrderedDict((n, self.flat[o:o + k].view(s)) for n, (o, k, s) in self.offsets.items())
        self.g = OrderedDict((n, self.grad[o:o + k].view(s)) for n, (o, k, s) in self.offsets.items())
        if self.flat.is_cuda:
            # filters that are views of this buffer keep persistent bf16 planes (hip.filter_split), dropped with the scope
            import weakref
            from . import hip
            weakref.finalize(self, hip.release_param_buffer, hip.register_param_buffer(self.flat))

    def names(self):
        return list(self.offsets.keys())


class ParamStore(object):
    """All variables of one model instance (generator + discriminator + non-trainables)."""

    def __init__(self, block_type='Pix2Pix', vocab_size=58, img=192, device='cuda', seed=0):
        if block_type == 'Pix2Pix':
            g, d, nt = pix2pix_param_specs(vocab_size, img)
        elif block_type == 'Residual':
            g = residual_generator_specs('fg', vocab_size, img)
            d, nt = residual_discriminator_specs()
        elif block_type == 'MRU':
            g = mru_generator_specs(vocab_size, img)
            d, nt = mru_discriminator_specs()
        elif block_type == 'BG':            # Background_Colorization (BASELINE config 5): residual generator + discriminator
            g, d, nt = residual_generator_specs('bg', vocab_size, img), bg_discriminator_specs(), []
        else:
            raise NotImplementedError('block_type %r: Pix2Pix (train+infer) and MRU/Residual/BG (generator forward) '
                                      'are built so far' % block_type)
        self.block_type = block_type
        self.device = device
        self.generator = Scope('generator', g, device)
        self.discriminator = Scope('discriminator', d, device)
        self.nontrainable = OrderedDict((n, torch.zeros(s, dtype=torch.float32, device=device)) for n, s, _ in nt)
        self._nt_specs = nt
        self.initialize(seed)

    def scope(self, name):
        return self.generator if name == 'generator' else self.discriminator

    def initialize(self, seed=0):
        """Reference initialisers, identical on every rank for one seed."""
        gen = torch.Generator().manual_seed(seed)
        for sc in (self.generator, self.discriminator):
            for n, shape, init in sc.specs:
                sc.p[n].copy_(_init_tensor(shape, init, gen))
            sc.adam_v.zero_()
            sc.adam_t = 0
        for n, shape, init in self._nt_specs:
            self.nontrainable[n].copy_(_init_tensor(shape, init, gen))
        self._planes_follow()

    def __getitem__(self, name):
        if name in self.nontrainable:
            return self.nontrainable[name]
        return self.scope(name.split('/', 1)[0]).p[name]

    def grad(self, name):
        return self.scope(name.split('/', 1)[0]).g[name]

    def names(self):
        return self.generator.names() + self.discriminator.names() + list(self.nontrainable.keys())

    def _planes_follow(self):
        """The bf16 planes of this store's filters follow weights written through torch NOW (not at the next launch Python
        happens to issue: a replayed hipGraph holds the planes' addresses and never looks at a tensor's version)."""
        if self.generator.flat.is_cuda:
            from . import hip
            hip.resplit_stale()

    def load_dict(self, d):
        """Copy values from a {tf_name: tensor/ndarray} mapping (checkpoints, test fixtures)."""
        for n in self.names():
            if n in d:
                self[n].copy_(torch.as_tensor(d[n], dtype=torch.float32).reshape(self[n].shape))
        self._planes_follow()

    def state_dict(self):
        out = OrderedDict((n, self[n].detach().cpu()) for n in self.names())
        for sc in (self.generator, self.discriminator):
            out['__adam_v__/' + sc.name] = sc.adam_v.detach().cpu()
            out['__adam_t__/' + sc.name] = torch.tensor(sc.adam_t)
            if sc.adam_m is not None:           # second optimizer slot (Adam beta1 != 0, RMSProp, AdaDelta)
                out['__adam_m__/' + sc.name] = sc.adam_m.detach().cpu()
        return out

    def load_state_dict(self, sd):
        self.load_dict(sd)
        for sc in (self.generator, self.discriminator):
            if '__adam_v__/' + sc.name in sd:
                sc.adam_v.copy_(sd['__adam_v__/' + sc.name])
                sc.adam_t = int(sd['__adam_t__/' + sc.name])
            if sc.adam_m is not None and '__adam_m__/' + sc.name in sd:
                sc.adam_m.copy_(sd['__adam_m__/' + sc.name])

    def parameter_count(self, scope):
        return sum(k for _, (o, k, s) in self.scope(scope).offsets.items())


class Buffers(object):
    """Named activation buffers, allocated once per (name, shape) and never freed, so device pointers
    recorded in a captured hipGraph stay valid while other shapes (caption lengths, batch sizes) come and go."""

    def __init__(self, device='cuda'):
        self.device = device
        self._b = {}

    def get(self, name, shape, dtype=torch.float32, zero_on_alloc=False):
        key = (name, tuple(int(s) for s in shape), dtype)
        t = self._b.get(key)
        if t is None:
            t = (torch.zeros if zero_on_alloc else torch.empty)(key[1], dtype=dtype, device=self.device)
            self._b[key] = t
        return t
