"""MRU generator (the reference's default ``--block_type MRU``) on the HIP kernels, forward / inference.

Reference: models_collection.py:68-147 (image_encoder_mru), :251-377 (generate_mru); mru.py:353-461
(mru_conv_block_v3), :527-591 (mru_deconv_block_v2), NUM_BLOCKS = 1.  This is the path the released pipeline runs
(Pipeline_utils/fg_color_utils.py:258-265 calls build_single_graph with the default block type).

Layout NHWC fp32.  Unlike the Pix2Pix / Residual blocks the MRU norms are *conditional* (per-sample scale/offset
rows selected by the class label, models_collection.py:29-34) and the activation is miu_relu, so the normalised
tensors are materialised by the pointwise kernels of csrc/mru_ops.hip instead of being folded into the consumer's
tile loads; every 3x3 / 7x7 / 1x1 contraction runs on the implicit-GEMM MFMA kernel with bias (+ lrelu for the
gates) in its epilogue.  Fusions that change no arithmetic:
  * channel concats ([miu(cbn(ht)) | x], [up(ht) | z | skip], [rg*up(ht) | z | skip]) are written once by
    ``ssc_concat_parts`` (norm + activation + nearest-2x upsample + gate folded into that single pass);
  * the 1x1 projection of the upsampled state is computed at low resolution (a pointwise conv commutes with
    nearest upsampling; its batch statistics are those of the upsampled tensor) and upsampled inside the blend;
  * ``ht_orig + h_new`` is the accumulate epilogue of the projection conv.
"""
import torch

from . import hip
from .hip import ACT_MIU, ACT_NONE, View
from .text_fusion import TextFusion

ENC_UNITS = [(1, 8, 64), (2, 64, 128), (3, 128, 256), (4, 256, 512)]                 # (unit, C_h, D); inp = 3 ch
DEC_UNITS = [(0, 512, 384), (2, 384, 256), (4, 256, 128), (6, 128, 128), (8, 128, 64)]      # (unit_num, C_h, D)


def _rows(t):
    return t.view(-1, t.shape[-1])


def _pad4(c):
    return (c + 3) // 4 * 4


class MRUGenerator(object):
    def __init__(self, store, bufs, lstm_hybrid=True):
        self.s, self.b = store, bufs
        self.lstm_hybrid = bool(lstm_hybrid)
        self.text = TextFusion(store, bufs)

    # ------------------------------------------------------------------ helpers
    def _const(self, c, value):
        t = self.b._b.get(('const', c, value))
        if t is None:
            t = torch.full((c,), float(value), dtype=torch.float32, device=self.b.device)
            self.b._b[('const', c, value)] = t
        return t

    def _cbn(self, tag, pre, raw, labels):
        """Conditional batch norm folded to per-sample (a, b): abn [N, 2C]."""
        s, B = self.s, self.b
        N, C = raw.shape[0], raw.shape[-1]
        ab = B.get(tag + '/' + pre + '/ab0', (2 * C,))
        st = B.get(tag + '/' + pre + '/st', (2 * C,))
        hip.bn_stats(_rows(raw), self._const(C, 1.0), self._const(C, 0.0), ab, st)
        abn = B.get(tag + '/' + pre + '/abn', (N, 2 * C))
        hip.call('ssc_cbn_fold', st, s[pre + '/scale'], s[pre + '/offset'], labels, N, C, abn)
        return abn

    def _norm_act(self, tag, pre, raw, labels, name):
        """miu_relu(cond_batchnorm(raw)) materialised."""
        abn = self._cbn(tag, pre, raw, labels)
        out = self.b.get(tag + '/' + pre + '/' + name, raw.shape)
        hip.concat_parts(out, [dict(x=raw, ab=abn, act=ACT_MIU)])
        return out

    def _conv(self, tag, pre, xv, cout, name, k_same=True, stride=1, epi=0, accumulate_into=None):
        s = self.s
        w = s[pre + '/weights']
        if accumulate_into is not None:
            out = accumulate_into
        else:
            out = self.b.get(tag + '/' + pre + '/' + name, (xv.N, -(-xv.H // stride), -(-xv.W // stride), cout))
        hip.conv_forward(xv, w, stride, 0, out, bias=s[pre + '/biases'], epi=epi, same=True,
                         accumulate=accumulate_into is not None)
        return out

    def _minmax(self, tag, pre, x, name):
        mm = self.b.get(tag + '/' + pre + '/' + name, (x.shape[0], 2, x.shape[-1]))
        hip.minmax_hw(x, mm)
        return mm

    def _pool(self, tag, name, x):
        N, H, W, C = x.shape
        out = self.b.get(tag + '/' + name, (N, H // 2, W // 2, C))
        hip.call('ssc_mean_pool2', x, C, out, C, N, H, W, C)
        return out

    # ------------------------------------------------------------------ blocks
    def _conv_block(self, tag, pre, xin, ht, d, labels):
        """mru_conv_block_v3 (stride 2): returns the mean-pooled new state."""
        B = self.b
        N, h, w, ch = ht.shape
        abn_in = self._cbn(tag, pre + '/norm_activation_in', ht, labels)
        full = B.get(tag + '/' + pre + '/full', (N, h, w, ch + 4), zero_on_alloc=True)
        hip.concat_parts(full, [dict(x=ht, ab=abn_in, act=ACT_MIU), dict(x=xin, C=3)])
        rg = self._conv(tag, pre + '/update_gate', View(full), ch, 'rg', epi=2)
        mm = self._minmax(tag, pre, rg, 'rg_mm')
        img = self._conv(tag, pre + '/Conv', View(xin), ch, 'img')
        htp = B.get(tag + '/' + pre + '/ht_plus', ht.shape)
        hip.call('ssc_mru_gate_merge', ht, rg, mm, img, htp, N, h * w, ch)
        hin = self._norm_act(tag, pre + '/norm_activation_merge_1', htp, labels, 'y')
        h1 = self._conv(tag, pre + '/Conv_1', View(hin), d, 'raw')
        h1a = self._norm_act(tag, pre + '/Conv_1', h1, labels, 'y')
        out = self._conv(tag, pre + '/Conv_2', View(h1a), d, 'raw')
        if ch != d:
            self._conv(tag, pre + '/Conv_3', View(ht), d, 'raw', accumulate_into=out)
        else:
            hip.call('ssc_axpy', out, ht, 1.0, out.numel())
        return self._pool(tag, pre + '/pooled', out)

    def _deconv_block(self, tag, pre, z, skip, ht, d, labels):
        """mru_deconv_block_v2 (stride 2): ht [N,h,w,C_h] -> [N,2h,2w,d]."""
        B = self.b
        N, h, w, ch = ht.shape
        H, W = 2 * h, 2 * w
        inp = [dict(x=z, C=3)] + ([dict(x=skip)] if skip is not None else [])
        ct = ch + 3 + (skip.shape[-1] if skip is not None else 0)
        full = B.get(tag + '/' + pre + '/full', (N, H, W, _pad4(ct)), zero_on_alloc=True)
        hip.concat_parts(full, [dict(x=ht, upsample=True)] + inp)
        rg = self._conv(tag, pre + '/Conv', View(full), ch, 'raw', epi=2)
        mm_r = self._minmax(tag, pre, rg, 'rg_mm')
        zg = self._conv(tag, pre + '/Conv_1', View(full), d, 'raw', epi=2)
        mm_z = self._minmax(tag, pre, zg, 'zg_mm')
        in2 = B.get(tag + '/' + pre + '/in2', (N, H, W, _pad4(ct)), zero_on_alloc=True)
        hip.concat_parts(in2, [dict(x=ht, upsample=True, gate=(rg, mm_r))] + inp)
        h1 = self._conv(tag, pre + '/Conv_2', View(in2), d, 'raw')
        h1a = self._norm_act(tag, pre + '/Conv_2', h1, labels, 'y')
        h2 = self._conv(tag, pre + '/Conv_3', View(h1a), d, 'raw')
        abn2 = self._cbn(tag, pre + '/Conv_3', h2, labels)
        out = B.get(tag + '/' + pre + '/out', (N, H, W, d))
        if ch != d:
            pj = self._conv(tag, pre + '/Conv_4', View(ht), d, 'raw')        # at low resolution (see module doc)
            abnp = self._cbn(tag, pre + '/Conv_4', pj, labels)
            hip.call('ssc_mru_blend', pj, abnp, 1, h2, abn2, zg, mm_z, out, N, H, W, d)
        else:
            hip.call('ssc_mru_blend', ht, None, 1, h2, abn2, zg, mm_z, out, N, H, W, d)
        return out

    # ------------------------------------------------------------------ forward
    def forward(self, sketches, text, labels, noise_vec, tag='g'):
        """sketches NCHW [N,3,H,W] (device), text int [N,T] (host), labels int32 [N] (device) = class ids,
        noise_vec [N,256] (device).  ctx['out'] = tanh image NHWC4."""
        s, B = self.s, self.b
        N, _, H, W = sketches.shape
        assert H % 32 == 0 and W % 32 == 0
        labels = labels.to(device=sketches.device, dtype=torch.int32).contiguous()
        xs = B.get(tag + '/xs', (N, H, W, 4), zero_on_alloc=True)
        hip.nchw_to_nhwc(sketches, xs, 0)
        pyr = [xs]          # mean-pool pyramid == AREA resize for the integer factors (models_collection.py:76-80, 264-267)
        for k in range(1, 5):
            pyr.append(self._pool(tag, 'pyr%d' % k, pyr[-1]))
        h0 = self._conv(tag, 'generator/Conv', View(xs), 8, 'raw', stride=2)
        enc = [h0]
        ht = h0
        for (u, ch, d), xin in zip(ENC_UNITS, pyr[1:5]):
            ht = self._conv_block(tag, 'generator/mru_conv_unit_t_%d_layer_0' % u, xin, ht, d, labels)
            if u == 4:      # last_unit (mru.py:651-653)
                ht = self._norm_act(tag, 'generator/mru_conv_unit_last_norm', ht, labels, 'y')
            enc.append(ht)
        e5 = enc[-1]
        ctx = {'tag': tag, 'N': N, 'H': H, 'W': W, 'enc': enc}
        if self.lstm_hybrid:
            feat, tctx = self.text.forward(e5, None, text, tag)
            ctx['tctx'] = tctx
        else:
            feat = e5
        hh, ww = e5.shape[1] * 2, e5.shape[2] * 2
        P = hh * ww
        pre = B.get(tag + '/noise_pre', (N, 64 * P))
        hip.matmul(noise_vec, s['generator/fully_connected/weights'], pre, bias=s['generator/fully_connected/biases'])
        noise = B.get(tag + '/noise', (N, hh, ww, 64))
        hip.call('ssc_miu_permute_fwd', pre, N, 64, P, noise)
        skips = [noise, enc[-3], enc[-4], enc[-5], None]
        zs = [pyr[4], pyr[3], pyr[2], pyr[1], pyr[0]]
        ht = feat
        dec = []
        for (u, ch, d), z, skip in zip(DEC_UNITS, zs, skips):
            assert ht.shape[-1] == ch
            ht = self._deconv_block(tag, 'generator/mru_deconv_unit_t_%d_layer_0' % u, z, skip, ht, d, labels)
            dec.append(ht)
        out = B.get(tag + '/gen', (N, H, W, 4))
        hip.conv_forward(View(ht), s['generator/Conv_1/weights'], 1, 0, out, nstore=4,
                         bias=s['generator/Conv_1/biases'], epi=1, same=True)
        ctx.update(out=out, out_coff=0, feat=feat, dec=dec, noise=noise)
        return ctx

    def output_nchw(self, ctx):
        N, H, W = ctx['N'], ctx['H'], ctx['W']
        o = torch.empty((N, 3, H, W), dtype=torch.float32, device=ctx['out'].device)
        hip.nhwc_to_nchw(ctx['out'], o, 0)
        return o


class MRUTower(object):
    """Inference tower for ``--block_type MRU`` (generator forward only; the MRU discriminator and the backward
    passes are not built yet)."""

    def __init__(self, img=192, vocab_size=58, device='cuda', seed=0, lstm_hybrid=True, **_):
        from .params import Buffers, ParamStore
        hip.lib()
        self.store = ParamStore('MRU', vocab_size, img, device, seed)
        self.bufs = Buffers(device)
        self.G = MRUGenerator(self.store, self.bufs, lstm_hybrid)

    def generate(self, sketches, text, noise_vec, labels=None):
        if labels is None:
            raise ValueError('the MRU generator is class-conditional: pass the class ids (image_data_class_id)')
        ctx = self.G.forward(sketches, text, labels, noise_vec, 'g')
        return self.G.output_nchw(ctx)

    def _no_training(self, *a, **k):
        raise NotImplementedError('--block_type MRU: only the generator forward (inference/test/validation) is '
                                  'built; train with --block_type Pix2Pix')

    d_gradients = g_gradients = train_iteration = _no_training
