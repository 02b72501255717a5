"""Bottleneck-residual generators on the HIP kernels (forward / inference).

Two reference graphs share every block here:
  * FG ``generate_residual``          models_collection.py:541-672 (blocks residual_util.py:81-171), NCHW API,
    selected by ``--block_type Residual``;
  * BG ``create_residual_generator``  bg_colorization_main.py:302-420 (blocks :217-299), NHWC API, 768x768,
    1024-channel bottleneck, 'mLSTM_G' caption cells, 3-channel region-segmentation branch (BASELINE config 5).

Layout follows pix2pix.py: NHWC fp32, every conv output stored RAW, the batch-statistics norm folded into a
per-channel (a, b) pair that the *consumer's* tile loads apply together with its relu/lrelu.  A bottleneck
therefore writes its three conv outputs once each and reads them once each; only the block output
``act(norm(block_3) + shortcut)`` is materialised (``ssc_residual_merge``), because it fans out to three
consumers (block_1, the identity/projection shortcut and a skip concat).  3-channel tensors are padded to 4.
"""
import torch

from . import hip
from .hip import ACT_LRELU, ACT_NONE, ACT_RELU, ACT_TANH, View
from .params import RESIDUAL_UNITS
from .text_fusion import TextFusion


def _rows(t):
    return t.view(-1, t.shape[-1])


class _Val(object):
    """A logical activation = act(a*t + b): raw tensor + folded norm + activation, applied on load."""

    def __init__(self, t, ab=None, act=ACT_NONE):
        self.t, self.ab, self.act = t, ab, act


def _view(a, b=None):
    if b is None:
        return View(a.t, None, a.ab, a.act)
    return View(a.t, b.t, a.ab, a.act, b.ab, b.act if b.act != a.act else -1)


class ResidualGenerator(object):
    def __init__(self, store, bufs, kind='fg', lstm_hybrid=True, size=64, seg_classes=3):
        assert kind in ('fg', 'bg')
        self.s, self.b, self.kind = store, bufs, kind
        self.fg = kind == 'fg'
        self.size, self.seg = size, seg_classes
        self.lstm_hybrid = bool(lstm_hybrid) or not self.fg
        self.text = TextFusion(store, bufs, 'generator/TextLSTM' if self.fg else 'generator/mLSTM_G')
        self.top = size * 8 if self.fg else size * 16

    # ------------------------------------------------------------------ pieces
    def _bn(self, tag, pre, raw, c_real=None):
        """Batch statistics of ``raw`` folded with scale/offset -> ab [2*C].  c_real < C: the tensor is channel
        padded and the affine of the pad channels is (0, 0)."""
        s, B = self.s, self.b
        C = raw.shape[-1]
        scale, offset = s[pre + '/scale'], s[pre + '/offset']
        if c_real is not None and c_real != C:
            sp = B.get(tag + '/' + pre + '/scale_p', (C,), zero_on_alloc=True)
            op = B.get(tag + '/' + pre + '/offset_p', (C,), zero_on_alloc=True)
            sp[:c_real].copy_(scale)
            op[:c_real].copy_(offset)
            scale, offset = sp, op
        ab = B.get(tag + '/' + pre + '/ab', (2 * C,))
        st = B.get(tag + '/' + pre + '/st', (2 * C,))
        hip.bn_stats(_rows(raw), scale, offset, ab, st)
        return ab

    def _tail(self, tag, pre, r1, c4, cout, act):
        """block_2 (3x3 SAME) and block_3 (1x1) of every bottleneck; r1 = raw block_1 output."""
        s, B = self.s, self.b
        N, h, w, _ = r1.shape
        ab1 = self._bn(tag, pre + '/block_1/batchnorm', r1)
        r2 = B.get(tag + '/' + pre + '/r2', (N, h, w, c4))
        hip.conv_forward(View(r1, None, ab1, act), s[pre + '/block_2/conv_ex/filter'], 1, 0, r2, same=True)
        ab2 = self._bn(tag, pre + '/block_2/batchnorm', r2)
        r3 = B.get(tag + '/' + pre + '/r3', (N, h, w, cout))
        hip.conv_forward(View(r2, None, ab2, act), s[pre + '/block_3/conv_ex/filter'], 1, 0, r3, same=True)
        ab3 = self._bn(tag, pre + '/block_3/batchnorm', r3)
        return r3, ab3

    def _merge(self, tag, pre, r3, ab3, sc, absc, act):
        out = self.b.get(tag + '/' + pre + '/out', r3.shape)
        M, C = _rows(r3).shape
        hip.call('ssc_residual_merge', r3, ab3, sc, absc, act, out, M, C)
        return _Val(out)

    def _en(self, tag, pre, xv, cout):
        """bottleneck_residual_en(stride=2), residual_util.py:81-109."""
        s, B = self.s, self.b
        c4 = cout // 4
        N, h, w = xv.N, xv.H // 2, xv.W // 2
        r1 = B.get(tag + '/' + pre + '/r1', (N, h, w, c4))
        hip.conv_forward(xv, s[pre + '/block_1/conv/filter'], 2, 1, r1)
        r3, ab3 = self._tail(tag, pre, r1, c4, cout, ACT_LRELU)
        sc = B.get(tag + '/' + pre + '/sc', (N, h, w, cout))
        hip.conv_forward(xv, s[pre + '/block_add/conv/filter'], 2, 1, sc)
        absc = self._bn(tag, pre + '/block_add/batchnorm', sc)
        return self._merge(tag, pre, r3, ab3, sc, absc, ACT_LRELU)

    def _de(self, tag, pre, xv, cout):
        """bottleneck_residual_de, residual_util.py:112-146."""
        s, B = self.s, self.b
        c4 = cout // 4
        N, h, w = xv.N, xv.H * 2, xv.W * 2
        r1 = B.get(tag + '/' + pre + '/r1', (N, h, w, c4))
        hip.deconv_forward(xv, s[pre + '/block_1/deconv/filter'], r1)
        r3, ab3 = self._tail(tag, pre, r1, c4, cout, ACT_RELU)
        sc = B.get(tag + '/' + pre + '/sc', (N, h, w, cout))
        hip.deconv_forward(xv, s[pre + '/block_add/deconv/filter'], sc)
        absc = self._bn(tag, pre + '/block_add/batchnorm', sc)
        return self._merge(tag, pre, r3, ab3, sc, absc, ACT_RELU)

    def _pu(self, tag, pre, x, act):
        """bottleneck_residual_pu, residual_util.py:149-171 (x is a materialised block output)."""
        s, B = self.s, self.b
        N, h, w, c = x.t.shape
        c4 = c // 4
        r1 = B.get(tag + '/' + pre + '/r1', (N, h, w, c4))
        hip.conv_forward(_view(x), s[pre + '/block_1/conv_ex/filter'], 1, 0, r1, same=True)
        r3, ab3 = self._tail(tag, pre, r1, c4, c, act)
        return self._merge(tag, pre, r3, ab3, x.t, None, act)

    def _pad3(self, tag, name, x_nhwc3):
        N, H, W, _ = x_nhwc3.shape
        xs = self.b.get(tag + '/' + name, (N, H, W, 4), zero_on_alloc=True)
        hip.call('ssc_affine_act', x_nhwc3, 3, None, 0, ACT_NONE, xs, 4, N * H * W, 3)
        return xs

    # ------------------------------------------------------------------ forward
    def forward(self, inputs, text, noise_vec=None, tag='g'):
        """FG: inputs NCHW [N,3,H,W], noise_vec [N,256] -> ctx['out'] NHWC4 (see ``output_nchw``).
        BG: inputs NHWC [N,H,W,3] -> ctx['image'], ctx['region_logits'] NHWC [N,H,W,3].
        text int [N,T] on the host."""
        s, B = self.s, self.b
        size, top = self.size, self.top
        top_bn = (lambda pre: pre) if self.fg else (lambda pre: pre + '/batchnorm')
        if self.fg:
            N, _, H, W = inputs.shape
            xs = B.get(tag + '/xs', (N, H, W, 4), zero_on_alloc=True)
            hip.nchw_to_nhwc(inputs, xs, 0)
        else:
            N, H, W, _ = inputs.shape
            xs = self._pad3(tag, 'xs', inputs.contiguous())
        assert H % 32 == 0 and W % 32 == 0
        # encoder_1: conv 7x7 s2 SAME + norm + lrelu (applied by the consumers)
        e1 = B.get(tag + '/e1', (N, H // 2, W // 2, size))
        hip.conv_forward(View(xs), s['generator/encoder_1/conv_ex/filter'], 2, 0, e1, same=True)
        layers = [_Val(e1, self._bn(tag, top_bn('generator/encoder_1'), e1), ACT_LRELU)]
        enc_c = [size, size * 2, size * 4, size * 8, top]
        for k in range(2, 6):
            out = self._en(tag, 'generator/encoder_%d_0' % k, _view(layers[-1]), enc_c[k - 1])
            for u in range(1, RESIDUAL_UNITS[k - 2]):
                out = self._pu(tag, 'generator/encoder_%d_%d' % (k, u), out, ACT_LRELU)
            layers.append(out)
        e5 = layers[-1].t
        hh, ww = e5.shape[1], e5.shape[2]
        ctx = {'tag': tag, 'N': N, 'H': H, 'W': W}
        if self.lstm_hybrid:
            feat, tctx = self.text.forward(e5, None, text, tag)
            ctx['tctx'] = tctx
        else:
            feat = e5
        featv = _Val(feat)
        reg = None
        if self.fg:
            cd = top // 8
            P = hh * ww
            pre = B.get(tag + '/noise_pre', (N, cd * P))
            hip.matmul(noise_vec, s['generator/fully_connected/weights'], pre, bias=s['generator/fully_connected/biases'])
            noise = B.get(tag + '/noise', (N, hh, ww, cd))
            hip.call('ssc_miu_permute_fwd', pre, N, cd, P, noise)
            first = _view(featv, _Val(noise))
        else:
            first = _view(featv)
            # region_br_projection: 1x1 conv 1024 -> seg + norm + relu
            rp = B.get(tag + '/reg_p', (N, hh, ww, 4), zero_on_alloc=True)
            hip.conv_forward(_view(layers[-1]), s['generator/region_br_projection/conv_ex/filter'], 1, 0, rp, nstore=4,
                             same=True)
            reg = _Val(rp, self._bn(tag, 'generator/region_br_projection/batchnorm', rp, self.seg), ACT_RELU)
        dec_out = {5: size * 8, 4: size * 4, 3: size * 2, 2: size}
        n_enc = len(layers)
        for dl, k in enumerate((5, 4, 3, 2)):
            skip = n_enc - dl - 1
            xv = first if dl == 0 else _view(layers[-1], layers[skip])
            out = self._de(tag, 'generator/decoder_%d_0' % k, xv, dec_out[k])
            for u in range(1, RESIDUAL_UNITS[skip - 1]):
                out = self._pu(tag, 'generator/decoder_%d_%d' % (k, u), out, ACT_RELU)
            layers.append(out)
            if reg is not None:
                reg = self._region_up(tag, k, reg)
        # decoder_1: deconv(concat[decoder_2, encoder_1]) + norm + tanh
        d1 = B.get(tag + '/d1', (N, H, W, 4))
        hip.deconv_forward(_view(layers[-1], layers[0]), s['generator/decoder_1/deconv/filter'], d1, nstore=4)
        ab1 = self._bn(tag, top_bn('generator/decoder_1'), d1, 3)
        ctx['layers'] = layers
        if self.fg:
            out = B.get(tag + '/gen', (N, H, W, 4))
            hip.call('ssc_affine_act', d1, 4, ab1, 4, ACT_TANH, out, 4, N * H * W, 4)
            ctx.update(out=out, out_coff=0, feat=feat)
        else:
            reg = self._region_up(tag, 1, reg)
            image = torch.empty((N, H, W, 3), dtype=torch.float32, device=d1.device)
            hip.call('ssc_affine_act', d1, 4, ab1, 4, ACT_TANH, image, 3, N * H * W, 3)
            logits = torch.empty((N, H, W, self.seg), dtype=torch.float32, device=d1.device)
            hip.call('ssc_affine_act', reg.t, 4, reg.ab, 4, ACT_RELU, logits, self.seg, N * H * W, self.seg)
            ctx.update(image=image, region_logits=logits, feat=feat)
        return ctx

    def _region_up(self, tag, k, reg):
        """region_br_k: deconv seg->seg + norm + relu (bg_colorization_main.py:392-397, 411-416)."""
        N, h, w, _ = reg.t.shape
        r = self.b.get(tag + '/reg_%d' % k, (N, 2 * h, 2 * w, 4), zero_on_alloc=True)
        hip.deconv_forward(_view(reg), self.s['generator/region_br_%d/deconv/filter' % k], r, nstore=4)
        return _Val(r, self._bn(tag, 'generator/region_br_%d/batchnorm' % k, r, self.seg), ACT_RELU)

    def output_nchw(self, ctx):
        N, H, W = ctx['N'], ctx['H'], ctx['W']
        o = torch.empty((N, 3, H, W), dtype=torch.float32, device=ctx['out'].device)
        hip.nhwc_to_nchw(ctx['out'], o, 0)
        return o


class ResidualTower(object):
    """Inference tower for ``--block_type Residual``: variables + activation buffers + generator, with the part of
    the trainer interface that inference / test / validation use.  The Residual training path
    (``discriminate_residual`` and the backward passes) is not built yet and says so."""

    def __init__(self, img=192, vocab_size=58, device='cuda', seed=0, lstm_hybrid=True, **_):
        from .params import Buffers, ParamStore
        hip.lib()       # fail loudly when the HIP library is missing
        self.store = ParamStore('Residual', vocab_size, img, device, seed)
        self.bufs = Buffers(device)
        self.G = ResidualGenerator(self.store, self.bufs, 'fg', lstm_hybrid)

    def generate(self, sketches, text, noise_vec):
        ctx = self.G.forward(sketches, text, noise_vec, 'g')
        return self.G.output_nchw(ctx)

    def _no_training(self, *a, **k):
        raise NotImplementedError('--block_type Residual: only the generator forward (inference/test/validation) is '
                                  'built; train with --block_type Pix2Pix')

    d_gradients = g_gradients = train_iteration = _no_training
