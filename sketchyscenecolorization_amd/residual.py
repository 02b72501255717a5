"""Bottleneck-residual networks on the HIP kernels: generators (forward + hand-written backward) and the
Residual discriminator.

Reference graphs sharing every block here:
  * FG ``generate_residual`` / ``discriminate_residual``   models_collection.py:541-672, 844-893 (blocks
    residual_util.py:81-171), NCHW API, selected by ``--block_type Residual``;
  * BG ``create_residual_generator``  bg_colorization_main.py:302-420 (blocks :217-299), NHWC API, 768x768,
    1024-channel bottleneck, 'mLSTM_G' caption cells, 3-channel region-segmentation branch (BASELINE config 5;
    forward only).

Layout follows pix2pix.py: NHWC fp32, every conv output stored RAW, the batch-statistics norm folded into a
per-channel (a, b) pair that the *consumer's* tile loads apply together with its relu/lrelu.  A bottleneck
therefore writes its three conv outputs once each and reads them once each; only the block output
``act(norm(block_3) + shortcut)`` is materialised (``ssc_residual_merge``), because it fans out to three
consumers (block_1, the identity/projection shortcut and a skip concat).  3-channel tensors are padded to 4.

Backward: per block, dz = g_out * act'(out) once, then norm-backward / wgrad / dgrad of the three convs and of the
projection shortcut; gradients of a tensor with several consumers accumulate in one buffer through the
``accumulate`` epilogue of the dgrad kernels (no separate add passes).
"""
import torch

from . import hip
from .hip import _dev_env
from .hip import ACT_LRELU, ACT_NONE, ACT_RELU, ACT_TANH, View
from .params import RESIDUAL_UNITS
from .text_fusion import TextFusion


import os as _os

_BLOCK_OUT_FUSED = _dev_env('SSC_BLOCK_OUT_FUSED', '1') == '1'     # ssc_block_out_backward (A/B switch)


def _rows(t):
    return t.view(-1, t.shape[-1])


class _Val(object):
    """A logical activation = act(a*t + b): raw tensor + folded norm (+ its statistics) + activation, applied on
    load.  ``pre`` names the norm variables (for the backward pass), ``c_real`` the unpadded channel count."""

    def __init__(self, t, ab=None, act=ACT_NONE, st=None, pre=None, c_real=None):
        self.t, self.ab, self.act, self.st, self.pre, self.c_real = t, ab, act, st, pre, c_real


def _view(a, b=None):
    if b is None:
        return View(a.t, None, a.ab, a.act)
    return View(a.t, b.t, a.ab, a.act, b.ab, b.act if b.act != a.act else -1)


class _Bottlenecks(object):
    """bottleneck_residual_en / de / pu (residual_util.py:81-171): forward records + backward."""

    _gtag = ''

    def _init_blocks(self, store, bufs):
        self.s, self.b = store, bufs

    # ------------------------------------------------------------------ forward pieces
    def _bn_prep(self, tag, pre, C, c_real=None):
        """(scale, offset, ab [2*C], stats [2*C]) of the norm ``pre`` over a C-channel tensor: the ``bn=`` argument of the
        conv that produces the tensor (hip.conv_forward / deconv_forward fold the batch statistics in their epilogue).
        c_real < C: the tensor is channel padded and the affine of the pad channels is (0, 0)."""
        s, B = self.s, self.b
        scale, offset = s[pre + '/scale'], s[pre + '/offset']
        if c_real is not None and c_real != C:
            sp = B.get(tag + '/' + pre + '/scale_p', (C,), zero_on_alloc=True)
            op = B.get(tag + '/' + pre + '/offset_p', (C,), zero_on_alloc=True)
            sp[:c_real].copy_(scale)
            op[:c_real].copy_(offset)
            scale, offset = sp, op
        ab = B.get(tag + '/' + pre + '/ab', (2 * C,))
        st = B.get(tag + '/' + pre + '/st', (2 * C,))
        return scale, offset, ab, st

    def _tail(self, tag, pre, r1, bn1, c4, cout, act, rec):
        """block_2 (3x3 SAME) and block_3 (1x1) of every bottleneck; r1 = raw block_1 output, bn1 its folded norm."""
        s, B = self.s, self.b
        N, h, w, _ = r1.shape
        ab1, st1 = bn1[2], bn1[3]
        r2 = B.get(tag + '/' + pre + '/r2', (N, h, w, c4))
        bn2 = self._bn_prep(tag, pre + '/block_2/batchnorm', c4)
        hip.conv_forward(View(r1, None, ab1, act), s[pre + '/block_2/conv_ex/filter'], 1, 0, r2, same=True, bn=bn2)
        ab2, st2 = bn2[2], bn2[3]
        r3 = B.get(tag + '/' + pre + '/r3', (N, h, w, cout))
        bn3 = self._bn_prep(tag, pre + '/block_3/batchnorm', cout)
        hip.conv_forward(View(r2, None, ab2, act), s[pre + '/block_3/conv_ex/filter'], 1, 0, r3, same=True, bn=bn3)
        ab3, st3 = bn3[2], bn3[3]
        rec.update(r1=r1, ab1=ab1, st1=st1, r2=r2, ab2=ab2, st2=st2, r3=r3, ab3=ab3, st3=st3)
        return r3, ab3

    def _merge(self, tag, pre, r3, ab3, sc, absc, act, rec):
        out = self.b.get(tag + '/' + pre + '/out', r3.shape)
        M, C = _rows(r3).shape
        hip.call('ssc_residual_merge', r3, ab3, sc, absc, act, out, M, C)
        rec.update(out=out, act=act, pre=pre, tag=tag)
        self._tape.append(rec)
        return _Val(out)

    def _en(self, tag, pre, srcs, cout):
        """bottleneck_residual_en(stride=2), residual_util.py:81-109.  srcs: one or two _Val (channel concat)."""
        s, B = self.s, self.b
        xv = _view(*srcs)
        c4 = cout // 4
        N, h, w = xv.N, xv.H // 2, xv.W // 2
        rec = {'kind': 'en', 'srcs': srcs, 'xv': xv}
        r1 = B.get(tag + '/' + pre + '/r1', (N, h, w, c4))
        bn1 = self._bn_prep(tag, pre + '/block_1/batchnorm', c4)
        hip.conv_forward(xv, s[pre + '/block_1/conv/filter'], 2, 1, r1, bn=bn1)
        r3, ab3 = self._tail(tag, pre, r1, bn1, c4, cout, ACT_LRELU, rec)
        sc = B.get(tag + '/' + pre + '/sc', (N, h, w, cout))
        bnsc = self._bn_prep(tag, pre + '/block_add/batchnorm', cout)
        hip.conv_forward(xv, s[pre + '/block_add/conv/filter'], 2, 1, sc, bn=bnsc)
        absc, stsc = bnsc[2], bnsc[3]
        rec.update(sc=sc, absc=absc, stsc=stsc)
        return self._merge(tag, pre, r3, ab3, sc, absc, ACT_LRELU, rec)

    def _de(self, tag, pre, srcs, cout):
        """bottleneck_residual_de, residual_util.py:112-146."""
        s, B = self.s, self.b
        xv = _view(*srcs)
        c4 = cout // 4
        N, h, w = xv.N, xv.H * 2, xv.W * 2
        rec = {'kind': 'de', 'srcs': srcs, 'xv': xv}
        r1 = B.get(tag + '/' + pre + '/r1', (N, h, w, c4))
        bn1 = self._bn_prep(tag, pre + '/block_1/batchnorm', c4)
        hip.deconv_forward(xv, s[pre + '/block_1/deconv/filter'], r1, bn=bn1)
        r3, ab3 = self._tail(tag, pre, r1, bn1, c4, cout, ACT_RELU, rec)
        sc = B.get(tag + '/' + pre + '/sc', (N, h, w, cout))
        bnsc = self._bn_prep(tag, pre + '/block_add/batchnorm', cout)
        hip.deconv_forward(xv, s[pre + '/block_add/deconv/filter'], sc, bn=bnsc)
        absc, stsc = bnsc[2], bnsc[3]
        rec.update(sc=sc, absc=absc, stsc=stsc)
        return self._merge(tag, pre, r3, ab3, sc, absc, ACT_RELU, rec)

    def _pu(self, tag, pre, x, act):
        """bottleneck_residual_pu, residual_util.py:149-171 (x is a materialised block output)."""
        s, B = self.s, self.b
        N, h, w, c = x.t.shape
        c4 = c // 4
        rec = {'kind': 'pu', 'srcs': (x,), 'xv': _view(x)}
        r1 = B.get(tag + '/' + pre + '/r1', (N, h, w, c4))
        bn1 = self._bn_prep(tag, pre + '/block_1/batchnorm', c4)
        hip.conv_forward(rec['xv'], s[pre + '/block_1/conv_ex/filter'], 1, 0, r1, same=True, bn=bn1)
        r3, ab3 = self._tail(tag, pre, r1, bn1, c4, c, act, rec)
        return self._merge(tag, pre, r3, ab3, x.t, None, act, rec)

    # ------------------------------------------------------------------ backward pieces
    def _gslot(self, t):
        """Gradient buffer of tensor ``t`` for this backward pass: (buffer, accumulate?)."""
        key = t.data_ptr()
        if key in self._gdone:
            return self._gdone[key], True
        # (_gtag: two backward passes through ONE forward context that run on different streams -- BGTrainer's fake pair -- must
        # not share these buffers: the pass names its set)
        g = self.b.get('grad_of/%s%d' % (self._gtag, key), t.shape)
        self._gdone[key] = g
        return g, False

    def _gget(self, t):
        return self._gdone.get(t.data_ptr())

    def _sums(self, tag, pre, raw, ab, st):
        """Partial sums of this norm's backward, to be delivered by the epilogue of the launch that produces its incoming
        gradient (hip.BnBwdSums / ssc_conv_forward_bnbwd)."""
        x2d = _rows(raw)
        buf = self.b.get(tag + '/gb/' + pre + '/bnsums', (hip.BnBwdSums.rows_needed(x2d.shape[0]), 2 * x2d.shape[1]))
        return hip.BnBwdSums(x2d, ab, st, buf)

    def _bn_bwd(self, tag, pre, raw, ab, st, g, act, need_params, accumulate, c_real=None, sums=None):
        """Backward through act(norm(raw)): returns d raw; writes / adds the scale and offset gradients."""
        s, B = self.s, self.b
        C = raw.shape[-1]
        dx = B.get(tag + '/gb/' + pre + '/dx', raw.shape)
        ds = do = None
        direct = need_params and not accumulate and (c_real is None or c_real == C)
        if direct:
            ds, do = s.grad(pre + '/scale'), s.grad(pre + '/offset')
        elif need_params:
            tmp = B.get(tag + '/gb/' + pre + '/dso', (2, C))
            ds, do = tmp[0], tmp[1]
        hip.bn_act_backward(_rows(raw), ab, st, _rows(g), act, _rows(dx), dscale=ds, doffset=do, pre=sums)
        if need_params and not direct:
            cr = C if c_real is None else c_real
            if accumulate:
                hip.call('ssc_axpy', s.grad(pre + '/scale'), ds, 1.0, cr)
                hip.call('ssc_axpy', s.grad(pre + '/offset'), do, 1.0, cr)
            else:
                s.grad(pre + '/scale').copy_(ds[:cr])
                s.grad(pre + '/offset').copy_(do[:cr])
        return dx

    def _block_out_backward(self, rec, g_out, dz, need_params, accumulate):
        """Gradient w.r.t. the raw outputs of block_3 and (en / de blocks) of the projection shortcut from the gradient w.r.t. the
        block output; the scale / offset gradients of both norms."""
        s, B = self.s, self.b
        tag, pre, kind = rec['tag'], rec['pre'], rec['kind']
        r3 = rec['r3']
        M, C = _rows(r3).shape
        two = kind != 'pu'
        dxa = B.get(tag + '/gb/' + pre + '/block_3/batchnorm/dx', r3.shape)
        dxb = B.get(tag + '/gb/' + pre + '/block_add/batchnorm/dx', r3.shape) if two else None
        sites = [pre + '/block_3/batchnorm'] + ([pre + '/block_add/batchnorm'] if two else [])
        direct = need_params and not accumulate
        grads = []
        for site in sites:
            if direct:
                grads.append((s.grad(site + '/scale'), s.grad(site + '/offset')))
            elif need_params:
                tmp = B.get(tag + '/gb/' + site + '/dso', (2, C))
                grads.append((tmp[0], tmp[1]))
            else:
                grads.append((None, None))
        if not two:
            grads.append((None, None))
        coef = B.get(tag + '/gb/' + pre + '/block_out/coef', (3 * C,))
        ws = hip.workspace()
        hip.call('ssc_block_out_backward', rec['out'], g_out, M, C, rec['act'], r3, rec['ab3'], rec['st3'],
                 rec['sc'] if two else None, rec['absc'] if two else None, rec['stsc'] if two else None, dz, dxa, dxb,
                 grads[0][0], grads[0][1], grads[1][0], grads[1][1], coef, ws, ws.numel() * 4)
        if need_params and accumulate:
            for site, (ds, do) in zip(sites, grads):
                hip.call('ssc_axpy', s.grad(site + '/scale'), ds, 1.0, C)
                hip.call('ssc_axpy', s.grad(site + '/offset'), do, 1.0, C)
        return dxa, dxb

    def _block_backward(self, rec, g_out, need_params=True, accumulate=False, need_input=True, input_slice=None):
        """g_out: gradient w.r.t. the block output.  Adds the gradients of the block's input sources to their
        slots (``_gslot``).  input_slice=(n_off, nn, out, nstore): instead write the gradient of that channel range
        of a single-source input into ``out`` (discriminator layer_1 -> generated image)."""
        s, B = self.s, self.b
        tag, pre, act, kind = rec['tag'], rec['pre'], rec['act'], rec['kind']
        out = rec['out']
        dz = B.get(tag + '/gb/' + pre + '/dz', out.shape)
        acc = accumulate
        if _BLOCK_OUT_FUSED:
            # dz = g_out * act'(out) and, with that one dz, the norm backward of block_3 and (en / de) of the projection
            # shortcut: one partial-sum launch, one fold, one streaming launch (ssc_block_out_backward); dz itself is only
            # written for the identity shortcut of a pu block
            dr3, dsc = self._block_out_backward(rec, g_out, dz if kind == 'pu' else None, need_params, acc)
        else:
            hip.bn_act_backward(_rows(out), None, None, _rows(g_out), act, _rows(dz))      # act'(z) from the sign of out
            # block_3 (1x1), block_2 (3x3 SAME)
            dr3 = self._bn_bwd(tag, pre + '/block_3/batchnorm', rec['r3'], rec['ab3'], rec['st3'], dz, ACT_NONE,
                               need_params, acc)
            dsc = None
        x3 = View(rec['r2'], None, rec['ab2'], act)
        if need_params:
            hip.conv_wgrad(x3, View(dr3), s.grad(pre + '/block_3/conv_ex/filter'), 1, 0, accumulate=acc)
        g2 = B.get(tag + '/gb/' + pre + '/g2', rec['r2'].shape)
        sums2 = self._sums(tag, pre + '/block_2', rec['r2'], rec['ab2'], rec['st2'])
        hip.conv_dgrad(View(dr3), s[pre + '/block_3/conv_ex/filter'], 1, 0, g2, bnbwd=sums2.take(act))
        dr2 = self._bn_bwd(tag, pre + '/block_2/batchnorm', rec['r2'], rec['ab2'], rec['st2'], g2, act, need_params, acc,
                           sums=sums2)
        x2 = View(rec['r1'], None, rec['ab1'], act)
        if need_params:
            hip.conv_wgrad(x2, View(dr2), s.grad(pre + '/block_2/conv_ex/filter'), 1, 1, accumulate=acc)
        g1 = B.get(tag + '/gb/' + pre + '/g1', rec['r1'].shape)
        sums1 = self._sums(tag, pre + '/block_1', rec['r1'], rec['ab1'], rec['st1'])
        hip.conv_dgrad(View(dr2), s[pre + '/block_2/conv_ex/filter'], 1, 1, g1, bnbwd=sums1.take(act))
        dr1 = self._bn_bwd(tag, pre + '/block_1/batchnorm', rec['r1'], rec['ab1'], rec['st1'], g1, act, need_params, acc,
                           sums=sums1)
        xv = rec['xv']
        branches = [(dr1, 'block_1')]
        if kind != 'pu':
            if dsc is None:
                dsc = self._bn_bwd(tag, pre + '/block_add/batchnorm', rec['sc'], rec['absc'], rec['stsc'], dz, ACT_NONE,
                                   need_params, acc)
            branches.append((dsc, 'block_add'))
        op = {'en': 'conv', 'de': 'deconv', 'pu': 'conv_ex'}[kind]
        for dy, blk in branches:
            name = pre + '/%s/%s/filter' % (blk, op)
            if need_params:
                if kind == 'de':
                    hip.deconv_wgrad(xv, View(dy), s.grad(name), accumulate=acc)
                else:
                    hip.conv_wgrad(xv, View(dy), s.grad(name), 2 if kind == 'en' else 1, 1, accumulate=acc)
        if not need_input:
            return
        if kind == 'pu':
            # identity shortcut: g_in = dz + dgrad(block_1); dz is dead after the norm backward above
            x = rec['srcs'][0].t
            hip.conv_dgrad(View(dr1), s[pre + '/block_1/conv_ex/filter'], 1, 1, dz, accumulate=True)
            assert x.data_ptr() not in self._gdone, 'a pu-block input feeds only this block'
            self._gdone[x.data_ptr()] = dz
            return
        if input_slice is not None:
            n_off, nn, gout, nstore = input_slice
            for i, (dy, blk) in enumerate(branches):
                hip.conv_dgrad(View(dy), s[pre + '/%s/conv/filter' % blk], 2, 1, gout, n_off=n_off, nn=nn,
                               nstore=nstore, accumulate=i > 0)
            return
        off = 0
        for src in rec['srcs']:
            C = src.t.shape[-1]
            if src.pre != 'input':          # 'input' = the network input: no gradient wanted
                slot, a = self._gslot(src.t)
                for dy, blk in branches:
                    w = s[pre + '/%s/%s/filter' % (blk, op)]
                    if kind == 'de':
                        hip.deconv_dgrad(View(dy), w, slot, n_off=off, nn=C, accumulate=a)
                    else:
                        hip.conv_dgrad(View(dy), w, 2, 1, slot, n_off=off, nn=C, accumulate=a)
                    a = True
            off += C


class ResidualGenerator(_Bottlenecks):
    def __init__(self, store, bufs, kind='fg', lstm_hybrid=True, size=64, seg_classes=3):
        assert kind in ('fg', 'bg')
        self._init_blocks(store, bufs)
        self.kind = kind
        self.fg = kind == 'fg'
        self.size, self.seg = size, seg_classes
        self.lstm_hybrid = bool(lstm_hybrid) or not self.fg
        self.text_stream = None     # set by the trainer: side stream for the image-independent half of the caption branch
        self.text = TextFusion(store, bufs, 'generator/TextLSTM' if self.fg else 'generator/mLSTM_G')
        self.top = size * 8 if self.fg else size * 16
        self._tape, self._gdone = [], {}

    def _pad3(self, tag, name, x_nhwc3):
        N, H, W, _ = x_nhwc3.shape
        xs = self.b.get(tag + '/' + name, (N, H, W, 4), zero_on_alloc=True)
        hip.call('ssc_affine_act', x_nhwc3, 3, None, 0, ACT_NONE, xs, 4, N * H * W, 3)
        return xs

    # ------------------------------------------------------------------ forward
    def forward(self, inputs, text, noise_vec=None, tag='g', out=None, out_coff=0):
        """FG: inputs NCHW [N,3,H,W], noise_vec [N,256] -> tanh image written to ``out[..., out_coff:out_coff+3]``
        (NHWC; default a 4-channel buffer, see ``output_nchw``).
        BG: inputs NHWC [N,H,W,3] -> ctx['image'], ctx['region_logits'] NHWC [N,H,W,3].
        text int [N,T] on the host (or a TextFusion.prepare result)."""
        s, B = self.s, self.b
        size, top = self.size, self.top
        self._tape = []
        if self.lstm_hybrid and self.text_stream is not None and hip.PROFILE is None:
            # the caption's word LSTM does not see the image: start it next to the encoder
            text = self.text.start_words(text, None, tag, self.text_stream)
        top_bn = (lambda pre: pre) if self.fg else (lambda pre: pre + '/batchnorm')
        if self.fg and inputs.shape[3] == 4 and inputs.shape[1] != 3:     # NHWC4 from hip.sketch_preprocess_u8
            N, H, W, _ = inputs.shape
            xs = inputs
        elif self.fg:
            N, _, H, W = inputs.shape
            xs = B.get(tag + '/xs', (N, H, W, 4), zero_on_alloc=True)
            hip.nchw_to_nhwc(inputs, xs, 0)
        else:
            N, H, W, _ = inputs.shape
            xs = self._pad3(tag, 'xs', inputs.contiguous())
        assert H % 32 == 0 and W % 32 == 0
        # encoder_1: conv 7x7 s2 SAME + norm + lrelu (applied by the consumers)
        e1 = B.get(tag + '/e1', (N, H // 2, W // 2, size))
        bne1 = self._bn_prep(tag, top_bn('generator/encoder_1'), size)
        hip.conv_forward(View(xs), s['generator/encoder_1/conv_ex/filter'], 2, 0, e1, same=True, bn=bne1)
        ab_e1, st_e1 = bne1[2], bne1[3]
        layers = [_Val(e1, ab_e1, ACT_LRELU, st_e1, top_bn('generator/encoder_1'))]
        enc_c = [size, size * 2, size * 4, size * 8, top]
        for k in range(2, 6):
            o = self._en(tag, 'generator/encoder_%d_0' % k, (layers[-1],), enc_c[k - 1])
            for u in range(1, RESIDUAL_UNITS[k - 2]):
                o = self._pu(tag, 'generator/encoder_%d_%d' % (k, u), o, ACT_LRELU)
            layers.append(o)
        e5 = layers[-1].t
        hh, ww = e5.shape[1], e5.shape[2]
        ctx = {'tag': tag, 'N': N, 'H': H, 'W': W, 'xs': xs, 'noise_vec': noise_vec}
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
            first = (featv, _Val(noise))
            ctx.update(noise_pre=pre, noise=noise)
        else:
            first = (featv,)
            # region_br_projection: 1x1 conv 1024 -> seg + norm + relu
            rp = B.get(tag + '/reg_p', (N, hh, ww, 4), zero_on_alloc=True)
            bnp = self._bn_prep(tag, 'generator/region_br_projection/batchnorm', 4, self.seg)
            hip.conv_forward(_view(layers[-1]), s['generator/region_br_projection/conv_ex/filter'], 1, 0, rp, nstore=4,
                             same=True, bn=bnp)
            abp, stp = bnp[2], bnp[3]
            reg = _Val(rp, abp, ACT_RELU, stp, 'generator/region_br_projection/batchnorm')
            ctx['region'] = [reg]
        dec_out = {5: size * 8, 4: size * 4, 3: size * 2, 2: size}
        n_enc = len(layers)
        for dl, k in enumerate((5, 4, 3, 2)):
            skip = n_enc - dl - 1
            srcs = first if dl == 0 else (layers[-1], layers[skip])
            o = self._de(tag, 'generator/decoder_%d_0' % k, srcs, dec_out[k])
            for u in range(1, RESIDUAL_UNITS[skip - 1]):
                o = self._pu(tag, 'generator/decoder_%d_%d' % (k, u), o, ACT_RELU)
            layers.append(o)
            if reg is not None:
                reg = self._region_up(tag, k, reg)
                ctx['region'].append(reg)
        # decoder_1: deconv(concat[decoder_2, encoder_1]) + norm + tanh
        d1 = B.get(tag + '/d1', (N, H, W, 4))
        v1 = _view(layers[-1], layers[0])
        bnd1 = self._bn_prep(tag, top_bn('generator/decoder_1'), 4, 3)
        hip.deconv_forward(v1, s['generator/decoder_1/deconv/filter'], d1, nstore=4, bn=bnd1)
        ab1, st1 = bnd1[2], bnd1[3]
        ctx.update(layers=layers, feat=feat, d1=d1, ab_d1=ab1, st_d1=st1, v1=v1, tape=self._tape,
                   bn_d1=top_bn('generator/decoder_1'))
        if self.fg:
            if out is None:
                out = B.get(tag + '/gen', (N, H, W, 4), zero_on_alloc=True)
                out_coff = 0
            ldo = out.shape[3]
            hip.call('ssc_affine_act', d1, 4, ab1, 4, ACT_TANH, out.view(-1)[out_coff:], ldo, N * H * W, 3)
            ctx.update(out=out, out_coff=out_coff)
        else:
            reg = self._region_up(tag, 1, reg)
            ctx['region'].append(reg)
            image = torch.empty((N, H, W, 3), dtype=torch.float32, device=d1.device)
            hip.call('ssc_affine_act', d1, 4, ab1, 4, ACT_TANH, image, 3, N * H * W, 3)
            logits = torch.empty((N, H, W, self.seg), dtype=torch.float32, device=d1.device)
            hip.call('ssc_affine_act', reg.t, 4, reg.ab, 4, ACT_RELU, logits, self.seg, N * H * W, self.seg)
            ctx.update(image=image, region_logits=logits)
        return ctx

    def _region_up(self, tag, k, reg):
        """region_br_k: deconv seg->seg + norm + relu (bg_colorization_main.py:392-397, 411-416)."""
        N, h, w, _ = reg.t.shape
        r = self.b.get(tag + '/reg_%d' % k, (N, 2 * h, 2 * w, 4), zero_on_alloc=True)
        bnr = self._bn_prep(tag, 'generator/region_br_%d/batchnorm' % k, 4, self.seg)
        hip.deconv_forward(_view(reg), self.s['generator/region_br_%d/deconv/filter' % k], r, nstore=4, bn=bnr)
        ab, st = bnr[2], bnr[3]
        return _Val(r, ab, ACT_RELU, st, 'generator/region_br_%d' % k)

    def output_nchw(self, ctx):
        N, H, W = ctx['N'], ctx['H'], ctx['W']
        o = torch.empty((N, 3, H, W), dtype=torch.float32, device=ctx['out'].device)
        hip.nhwc_to_nchw(ctx['out'], o, ctx['out_coff'])
        return o

    # ------------------------------------------------------------------ backward (FG)
    def backward(self, ctx, dpre, on_section=None, dlogits=None):
        """dpre [N,H,W,4]: gradient w.r.t. the pre-tanh output (= norm(decoder_1 deconv)).  Writes every generator
        gradient; ``on_section(name)`` as in Pix2PixGenerator.backward ('decoders', 'text', 'encoders').
        BG generator: ``dlogits`` [N,H,W,4] = gradient w.r.t. the region-mask logits (region branch, :331-416)."""
        s, B = self.s, self.b
        done = on_section if on_section is not None else (lambda name: None)
        tag, N = ctx['tag'], ctx['N']
        layers, tape = ctx['layers'], ctx['tape']
        self._gdone = {}
        if not self.fg:
            self._region_backward(ctx, dlogits)
        # decoder_1
        dd1 = self._bn_bwd(tag, ctx['bn_d1'], ctx['d1'], ctx['ab_d1'], ctx['st_d1'], dpre, ACT_NONE, True,
                           False, c_real=3)
        f1 = s['generator/decoder_1/deconv/filter']
        hip.deconv_wgrad(ctx['v1'], View(dd1), s.grad('generator/decoder_1/deconv/filter'))
        off = 0
        for src in (layers[-1], layers[0]):
            C = src.t.shape[-1]
            slot, a = self._gslot(src.t)
            hip.deconv_dgrad(View(dd1), f1, slot, n_off=off, nn=C, accumulate=a)
            off += C
        # decoder stages 2..5 then encoder stages 5..2: the tape in reverse
        n_dec = sum(RESIDUAL_UNITS)
        for i, rec in enumerate(reversed(tape)):
            if i == n_dec:
                self._after_decoders(ctx, done)
            g_out = self._gget(rec['out'])
            assert g_out is not None, rec['pre']
            self._block_backward(rec, g_out)
        # encoder_1: norm + lrelu feed encoder_2_0 (two convs) and the decoder_1 skip, all through lrelu
        e1 = layers[0]
        de1 = self._bn_bwd(tag, e1.pre, e1.t, e1.ab, e1.st, self._gget(e1.t), ACT_LRELU, True, False)
        hip.conv_wgrad(View(ctx['xs']), View(de1), s.grad('generator/encoder_1/conv_ex/filter'), 2,
                       hip.same_pad_before(ctx['H'], 7, 2))
        done('encoders')

    def _region_backward(self, ctx, dlogits):
        """region_br_1 .. region_br_5 and the 1x1 projection (3-channel tensors padded to 4)."""
        s, B = self.s, self.b
        tag = ctx['tag']
        chain = ctx['region']           # [projection, br_5, br_4, br_3, br_2, br_1]
        g = dlogits
        for idx in range(len(chain) - 1, 0, -1):
            cur, prev = chain[idx], chain[idx - 1]
            dr = self._bn_bwd(tag, cur.pre + '/batchnorm', cur.t, cur.ab, cur.st, g, ACT_RELU, True, False, c_real=self.seg)
            name = cur.pre + '/deconv/filter'
            hip.deconv_wgrad(_view(prev), View(dr), s.grad(name))
            gp = B.get(tag + '/gb/' + cur.pre + '/gin', prev.t.shape, zero_on_alloc=True)
            hip.deconv_dgrad(View(dr), s[name], gp, n_off=0, nn=self.seg)
            g = gp
        proj = chain[0]
        drp = self._bn_bwd(tag, proj.pre, proj.t, proj.ab, proj.st, g, ACT_RELU, True, False, c_real=self.seg)
        e5 = ctx['layers'][4]
        w = s['generator/region_br_projection/conv_ex/filter']
        hip.conv_wgrad(_view(e5), View(drp), s.grad('generator/region_br_projection/conv_ex/filter'), 1, 0)
        slot, a = self._gslot(e5.t)
        hip.conv_dgrad(View(drp), w, 1, 0, slot, k_real=self.seg, accumulate=a)

    def _after_decoders(self, ctx, done):
        """Noise head and caption branch, between the decoder and encoder halves of the tape."""
        s, B = self.s, self.b
        tag, N = ctx['tag'], ctx['N']
        feat = ctx['feat']
        g_feat = self._gget(feat)
        if self.fg:
            noise = ctx['noise']
            g_noise = self._gget(noise)
            P = noise.shape[1] * noise.shape[2]
            cd = noise.shape[3]
            dpre_fc = B.get(tag + '/gb/noise_dpre', (N, cd * P))
            hip.call('ssc_miu_permute_bwd', ctx['noise_pre'], g_noise, N, cd, P, dpre_fc)
            hip.matmul_tn(ctx['noise_vec'], dpre_fc, s.grad('generator/fully_connected/weights'))
            hip.call('ssc_group_rowsum', dpre_fc, cd * P, 1, N, cd * P, s.grad('generator/fully_connected/biases'), 0)
        done('decoders')
        e5 = ctx['layers'][4].t
        if self.lstm_hybrid:
            dy5 = self.text.backward(ctx['tctx'], g_feat)
            if dy5 is None:
                ge5 = B.get(tag + '/gb/ge5', e5.shape)
                hip.fill(ge5, 0.0)
            else:
                ge5 = dy5.view(e5.shape)
            prev = self._gget(e5)           # BG: the region projection already contributed
            if prev is None:
                self._gdone[e5.data_ptr()] = ge5
            else:
                hip.call('ssc_axpy', prev, ge5, 1.0, ge5.numel())
        else:       # feat IS encoder_5's output: its slot already holds the gradient
            for nm in ('embedding', 'RNN/WLSTM/multi_rnn_cell/cell_0/basic_lstm_cell/kernel',
                       'RNN/WLSTM/multi_rnn_cell/cell_0/basic_lstm_cell/bias',
                       'RNN/ALSTM/multi_rnn_cell/cell_0/basic_lstm_cell/kernel',
                       'RNN/ALSTM/multi_rnn_cell/cell_0/basic_lstm_cell/bias'):
                hip.fill(s.grad(self.text.emb_name.rsplit('/', 1)[0] + '/' + nm), 0.0)
        done('text')


class BGDiscriminator(_Bottlenecks):
    """create_residual_discriminator (bg_colorization_main.py:550-580): five stride-2 encoder bottlenecks on
    concat[inputs, targets]; the sigmoid of the last block's output is the patch prediction [N, H/32, W/32, 1024]
    (the sigmoid lives in the loss kernel, ``forward`` returns the pre-sigmoid block output)."""

    def __init__(self, store, bufs, ndf=64):
        self._init_blocks(store, bufs)
        self.chans = [ndf, ndf * 2, ndf * 4, ndf * 8, 1024]
        self._tape, self._gdone = [], {}

    def forward(self, xd, tag):
        """xd NHWC [N,H,W,8] = [inputs(3), targets(3), 0, 0]."""
        self._tape = []
        cur = _Val(xd, pre='input')
        for k in range(1, 6):
            cur = self._en(tag, 'discriminator/layer_%d' % k, (cur,), self.chans[k - 1])
        return {'tag': tag, 'N': xd.shape[0], 'xd': xd, 'tape': self._tape, 'z': cur.t}

    def backward(self, ctx, dz, need_params, need_input, accumulate):
        """dz: gradient w.r.t. the pre-sigmoid output.  Returns d loss / d targets as NHWC [N,H,W,4] if need_input."""
        B = self.b
        self._gdone = {}
        self._gdone[ctx['z'].data_ptr()] = dz
        self._gtag = ctx['tag'] + '/'
        dgen = None
        tape = ctx['tape']
        for i, rec in enumerate(reversed(tape)):
            g_out = self._gget(rec['out'])
            if i == len(tape) - 1:
                if need_input:
                    xd = ctx['xd']
                    dgen = B.get(ctx['tag'] + '/gb/dgen', (ctx['N'], xd.shape[1], xd.shape[2], 4))
                    self._block_backward(rec, g_out, need_params, accumulate, True, input_slice=(3, 3, dgen, 4))
                else:
                    self._block_backward(rec, g_out, need_params, accumulate, False)
            else:
                self._block_backward(rec, g_out, need_params, accumulate, True)
        return dgen


class ResidualDiscriminator(_Bottlenecks):
    """discriminate_residual (models_collection.py:844-893): five stride-2 encoder bottlenecks on
    concat[sketch, target], a 4x4 s1 SAME patch head on layer_5 and the spectral-normed class head on the spatial
    mean of layer_4.  Same interface as Pix2PixDiscriminator."""

    def __init__(self, store, bufs, sn=True):
        self._init_blocks(store, bufs)
        self.sn = bool(sn)
        self.chans = [64, 128, 256, 512, 512]
        self._tape, self._gdone = [], {}

    def prepare_sn(self):
        s, B = self.s, self.b
        W = s['discriminator/fully_connected/weights']
        m, n = W.shape
        if not self.sn:
            return {'wbar': W}
        sn = {'v': B.get('d/sn/v', (m,)), 'u_new': B.get('d/sn/u_new', (1, n)), 'wbar': B.get('d/sn/wbar', (m, n)),
              'aux': B.get('d/sn/aux', (4,)), 'gwbar': B.get('d/sn/gwbar', (m, n)), 'n_acc': 0}
        hip.call('ssc_sn_forward', W, s['discriminator/fully_connected/u'], m, n, sn['v'], sn['u_new'], sn['wbar'],
                 sn['aux'])
        return sn

    def forward(self, xd, sn, tag):
        """xd NHWC [N,H,W,8] = [discrim_inputs(3), discrim_targets(3), 0, 0]."""
        s, B = self.s, self.b
        N = xd.shape[0]
        self._tape = []
        cur = _Val(xd, pre='input')
        outs = []
        for k in range(1, 6):
            cur = self._en(tag, 'discriminator/layer_%d' % k, (cur,), self.chans[k - 1])
            outs.append(cur)
        l4, l5 = outs[3].t, outs[4].t
        disc = B.get(tag + '/disc', (N, l5.shape[1], l5.shape[2], 4))
        hip.conv_forward(View(l5), s['discriminator/layer_5/conv_ex/filter'], 1, 0, disc, nstore=4, same=True)
        P4 = l4.shape[1] * l4.shape[2]
        img = B.get(tag + '/img', (N, 512))
        hip.call('ssc_act_mean_hw', l4, None, ACT_NONE, N, P4, 512, img)
        K = s['discriminator/fully_connected/weights'].shape[1]
        logits = B.get(tag + '/logits', (N, K))
        hip.call('ssc_fc_small_fwd', img, sn['wbar'], s['discriminator/fully_connected/biases'], N, 512, K, logits)
        return {'tag': tag, 'N': N, 'xd': xd, 'tape': self._tape, 'l4': l4, 'l5': l5, 'img': img, 'logits': logits,
                'disc': disc, 'P4': P4}

    def backward(self, ctx, dl5, dlogits, sn, need_params, need_input, accumulate):
        """dl5 [N,h5,w5,4] (channel 0 real), dlogits [N,K] or None; see Pix2PixDiscriminator.backward."""
        s, B = self.s, self.b
        tag, N, l4, l5 = ctx['tag'], ctx['N'], ctx['l4'], ctx['l5']
        self._gdone = {}
        w = s['discriminator/layer_5/conv_ex/filter']
        if need_params:
            hip.conv_wgrad(View(l5), View(dl5), s.grad('discriminator/layer_5/conv_ex/filter'), 1, 1,
                           accumulate=accumulate)
        g5, _ = self._gslot(l5)
        hip.conv_dgrad(View(dl5), w, 1, 1, g5, k_real=1)
        if dlogits is not None:
            dimg = B.get(tag + '/gb/dimg', (N, 512))
            K = dlogits.shape[1]
            gw, gb_, acc = None, None, 0
            if need_params:
                gb_ = s.grad('discriminator/fully_connected/biases')
                if self.sn:
                    gw, acc = sn['gwbar'], int(sn['n_acc'] > 0)
                    sn['n_acc'] += 1
                else:
                    gw, acc = s.grad('discriminator/fully_connected/weights'), int(accumulate)
            hip.call('ssc_fc_small_bwd', ctx['img'], sn['wbar'], dlogits, N, 512, K, dimg, gw, gb_, acc)
            g4, _ = self._gslot(l4)     # first contribution: mean over the P4 positions
            hip.fill(g4, 0.0)
            hip.call('ssc_add_row_bcast', g4, dimg, 1.0 / ctx['P4'], N, ctx['P4'], 512)
        dgen = None
        tape = ctx['tape']
        for i, rec in enumerate(reversed(tape)):
            first_layer = i == len(tape) - 1
            g_out = self._gget(rec['out'])
            if first_layer:
                if need_input:
                    xd = ctx['xd']
                    dgen = B.get(tag + '/gb/dgen', (N, xd.shape[1], xd.shape[2], 4))
                    self._block_backward(rec, g_out, need_params, accumulate, True, input_slice=(3, 3, dgen, 4))
                else:
                    self._block_backward(rec, g_out, need_params, accumulate, False)
            else:
                self._block_backward(rec, g_out, need_params, accumulate, True)
        return dgen

    def finish_sn_backward(self, sn, accumulate=False):
        s, B = self.s, self.b
        if not self.sn:
            return
        W = s['discriminator/fully_connected/weights']
        m, n = W.shape
        gW = s.grad('discriminator/fully_connected/weights')
        if sn['n_acc'] == 0:
            if not accumulate:
                hip.fill(gW, 0.0)
            return
        hip.call('ssc_sn_backward', W, s['discriminator/fully_connected/u'], sn['v'], sn['u_new'], sn['aux'],
                 sn['gwbar'], m, n, gW, int(accumulate), B.get('d/sn/scratch', (m,)))
