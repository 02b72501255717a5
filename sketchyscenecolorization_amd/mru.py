"""MRU generator and discriminator (the reference's default ``--block_type MRU``) on the HIP kernels:
forward and hand-written backward.

Reference: models_collection.py:68-147 (image_encoder_mru), :251-377 (generate_mru), :676-786 (discriminate_mru);
mru.py:353-461 (mru_conv_block_v3), :527-591 (mru_deconv_block_v2), NUM_BLOCKS = 1.  The generator forward is the path
the released pipeline runs (Pipeline_utils/fg_color_utils.py:258-265 calls build_single_graph with the default block).

Layout NHWC fp32.  Unlike the Pix2Pix / Residual blocks the generator's norms are *conditional* (per-sample
scale/offset rows selected by the class label, models_collection.py:29-34) and its activation is miu_relu; the
discriminator has no norm, a trainable-leak prelu and a spectral-normed weight in every conv.  Normalised /
activated tensors are therefore materialised by the pointwise kernels of csrc/mru_ops.hip instead of being folded
into the consumer's tile loads; every 3x3 / 7x7 / 1x1 contraction (and its two gradients) runs on the implicit-GEMM
MFMA kernel with bias (+ lrelu for the gates) in its epilogue.  Fusions that change no arithmetic:
  * channel concats ([act(ht) | x], [up(ht) | z | skip], [rg*up(ht) | z | skip]) are written once by
    ``ssc_concat_parts`` (norm + activation + nearest-2x upsample + gate folded into that single pass), and the
    gradient of the whole concat accumulates in one buffer through the dgrad kernels' accumulate epilogue;
  * the 1x1 projection of the upsampled state is computed at low resolution (a pointwise conv commutes with
    nearest upsampling; its batch statistics are those of the upsampled tensor) and upsampled inside the blend;
  * ``ht_orig + h_new`` is the accumulate epilogue of the projection conv.
"""
import torch

from . import hip
from .hip import _dev_env
from .hip import ACT_MIU, ACT_NONE, ACT_PRELU, View
from .text_fusion import TextFusion

import os as _os
_SPLIT_DGRAD = _dev_env('SSC_MRU_SPLIT_DGRAD', '1') == '1'
_FUSE_MINMAX = _dev_env('SSC_MRU_FUSE_MINMAX', '1') == '1'           # gate extrema out of the conv epilogue (A/B)
_FUSE_CBN_STATS = _dev_env('SSC_MRU_FUSE_STATS', '1') == '1'      # conditional-norm statistics out of the conv epilogue (A/B)

ENC_UNITS = [(1, 8, 64), (2, 64, 128), (3, 128, 256), (4, 256, 512)]                 # (unit, C_h, D); inp = 3 ch
DEC_UNITS = [(0, 512, 384), (2, 384, 256), (4, 256, 128), (6, 128, 128), (8, 128, 64)]      # (unit_num, C_h, D)
DISC_UNITS = [(1, 8, 128), (2, 128, 256), (3, 256, 512), (4, 512, 768)]
N_LABELS = 25
REG_CONV = 1e-5     # ly.l2_regularizer(weight_decay_rate=1e-5) on the block convs (mru.py:600, 664 -> :381, 545)


def _rows(t):
    return t.view(-1, t.shape[-1])


def _pad4(c):
    return (c + 3) // 4 * 4


class _MRUBlocks(object):
    """mru_conv_block_v3 forward / backward shared by the generator's encoder (conditional norm + miu_relu, plain
    weights) and the discriminator (prelu, spectral-normed weights)."""

    scope = 'generator'
    cond_norm = True

    def _init(self, store, bufs):
        self.s, self.b = store, bufs
        self._gdone = {}
        self._pre_stats = {}        # conv outputs whose batch statistics the conv's epilogue has already taken (-> scope)
        self.loss_acc = None        # device double scalar the regularisation terms are added to (set by the trainer)
        self._sn = None

    # ------------------------------------------------------------------ small helpers
    def _const(self, c, value):
        t = self.b._b.get(('const', c, value))
        if t is None:
            t = torch.full((c,), float(value), dtype=torch.float32, device=self.b.device)
            self.b._b[('const', c, value)] = t
        return t

    def _ws(self):
        return hip.workspace()

    def _w(self, pre):
        """Weights the convolution multiplies with: W (generator) or W / sigma (discriminator, sn.py)."""
        if self._sn is not None:
            return self._sn[pre]['wbar'].view(self.s[pre + '/weights'].shape)
        return self.s[pre + '/weights']

    def _gslot(self, t):
        key = t.data_ptr()
        if key in self._gdone:
            return self._gdone[key], True
        g = self.b.get('grad_of/%d' % key, t.shape)
        self._gdone[key] = g
        return g, False

    def _gget(self, t):
        return self._gdone.get(t.data_ptr())

    def _gset(self, t, g):
        self._gdone[t.data_ptr()] = g

    # ------------------------------------------------------------------ norm + activation
    def _na_fwd(self, tag, scope, raw, labels, name='y'):
        """norm_activ (mru.py:367-376): generator = miu_relu(cond_batchnorm(raw)); discriminator = prelu(raw)."""
        s, B = self.s, self.b
        out = B.get(tag + '/' + scope + '/' + name, raw.shape)
        if not self.cond_norm:
            hip.concat_parts(out, [dict(x=raw, ab=s[scope + '/prelu/param'].view(1), act=ACT_PRELU)])
            return out, None
        N, C = raw.shape[0], raw.shape[-1]
        ab = B.get(tag + '/' + scope + '/ab0', (2 * C,))
        st = B.get(tag + '/' + scope + '/st', (2 * C,))
        if self._pre_stats.pop(raw.data_ptr(), None) != scope:
            hip.bn_stats(_rows(raw), self._const(C, 1.0), self._const(C, 0.0), ab, st)
        abn = B.get(tag + '/' + scope + '/abn', (N, 2 * C))
        hip.call('ssc_cbn_fold', st, s[scope + '/scale'], s[scope + '/offset'], labels, N, C, abn)
        hip.concat_parts(out, [dict(x=raw, ab=abn, act=ACT_MIU)])
        return out, (abn, st)

    def _na_part(self, tag, scope, raw, labels):
        """norm_activ of ``raw`` as a PART of a concat (hip.concat_parts): the norm + activation are applied while the concat is
        written, the normalised tensor itself is never materialised (its only consumer is that concat)."""
        s, B = self.s, self.b
        if not self.cond_norm:
            return dict(x=raw, ab=s[scope + '/prelu/param'].view(1), act=ACT_PRELU), None
        abn, st = self._cbn(tag, scope, raw, labels)
        return dict(x=raw, ab=abn, act=ACT_MIU), (abn, st)

    def _cbn(self, tag, scope, raw, labels):
        s, B = self.s, self.b
        N, C = raw.shape[0], raw.shape[-1]
        ab = B.get(tag + '/' + scope + '/ab0', (2 * C,))
        st = B.get(tag + '/' + scope + '/st', (2 * C,))
        if self._pre_stats.pop(raw.data_ptr(), None) != scope:
            hip.bn_stats(_rows(raw), self._const(C, 1.0), self._const(C, 0.0), ab, st)
        abn = B.get(tag + '/' + scope + '/abn', (N, 2 * C))
        hip.call('ssc_cbn_fold', st, s[scope + '/scale'], s[scope + '/offset'], labels, N, C, abn)
        return abn, st

    def _na_bwd(self, tag, scope, raw, aux, labels, gy, ldg, dx, acc_dx, need_params, acc_params, act=ACT_MIU):
        """Backward of _na_fwd: dx (+)= d raw, parameter gradients written / added."""
        s = self.s
        N, C = raw.shape[0], raw.shape[-1]
        P = raw.numel() // (N * C)
        ws = self._ws()
        if not self.cond_norm:
            dleak = s.grad(scope + '/prelu/param').view(1) if need_params else self.b.get('scratch/dleak', (1,))
            hip.call('ssc_prelu_backward', raw, C, s[scope + '/prelu/param'].view(1), gy, ldg, N * P, C, dx,
                     dx.shape[-1], int(acc_dx), dleak, int(acc_params and need_params), ws, ws.numel() * 4)
            return
        abn, st = aux
        ds = s.grad(scope + '/scale') if need_params else None
        do = s.grad(scope + '/offset') if need_params else None
        hip.call('ssc_cbn_act_backward', raw, abn, st, s[scope + '/scale'], labels, N_LABELS, gy, ldg, act, N, P, C,
                 dx, dx.shape[-1], int(acc_dx), ds, do, int(acc_params), ws, ws.numel() * 4)

    # ------------------------------------------------------------------ conv with bias, its gradients
    def _conv(self, tag, pre, xv, cout, name, stride=1, epi=0, accumulate_into=None, nstore=None, stats=False, minmax=None):
        """stats=True (generator): the conv's output goes to the conditional norm of the same scope -- its batch statistics are
        taken by the conv's epilogue (hip.conv_forward(bn=...)) instead of a pass of their own over the tensor; ``_cbn`` /
        ``_na_fwd`` find them under the scope's buffer names."""
        s = self.s
        if accumulate_into is not None:
            out = accumulate_into
        else:
            co = cout if nstore is None else nstore
            out = self.b.get(tag + '/' + pre + '/' + name, (xv.N, -(-xv.H // stride), -(-xv.W // stride), co))
        bn = None
        if stats and _FUSE_CBN_STATS and self.cond_norm and accumulate_into is None and nstore is None and epi == 0:
            ab = self.b.get(tag + '/' + pre + '/ab0', (2 * cout,))
            st = self.b.get(tag + '/' + pre + '/st', (2 * cout,))
            bn = (self._const(cout, 1.0), self._const(cout, 0.0), ab, st)
            self._pre_stats[out.data_ptr()] = pre
        if minmax is not None and (accumulate_into is not None or nstore is not None or not _FUSE_MINMAX):
            hip.conv_forward(xv, self._w(pre), stride, 0, out, bias=s[pre + '/biases'], epi=epi, same=True,
                             accumulate=accumulate_into is not None, nstore=nstore)
            hip.minmax_hw(out, minmax)
            return out
        hip.conv_forward(xv, self._w(pre), stride, 0, out, bias=s[pre + '/biases'], epi=epi, same=True,
                         accumulate=accumulate_into is not None, nstore=nstore, bn=bn, minmax=minmax)
        return out

    def _conv_param_grads(self, pre, xv, dy, stride, need_params, acc, reg):
        """Filter + bias gradients of a SAME conv.  dy [N,h,w,Cout(+pad)] = gradient of its (pre-activation) output."""
        if not need_params:
            return
        s = self.s
        w = s[pre + '/weights']
        k = w.shape[0]
        if self._sn is not None:        # gradient w.r.t. W / sigma, summed over the calls of this step
            st = self._sn[pre]
            gw, a = st['gwbar'].view(w.shape), st['n_acc'] > 0
            st['n_acc'] += 1
        else:
            gw, a = s.grad(pre + '/weights'), acc
        hip.conv_wgrad(xv, View(dy), gw, stride, hip.same_pad_before(xv.H, k, stride), accumulate=a)
        if self._sn is None and reg and self.loss_acc is not None:
            hip.call('ssc_l2_reg', w, w.numel(), REG_CONV, self.loss_acc, gw)
        ws = self._ws()
        cb = s[pre + '/biases'].numel()
        hip.call('ssc_colsum', dy, dy.shape[-1], dy.numel() // dy.shape[-1], cb, s.grad(pre + '/biases'), int(acc), ws,
                 ws.numel() * 4)

    def _conv_dgrad(self, pre, dy, out, n_off=0, nn=None, nstore=None, accumulate=False, k_real=None):
        w = self._w(pre)
        k = w.shape[0]
        n_all = w.shape[2] - n_off if nn is None else nn
        n_st = n_all if nstore is None else nstore
        wide = n_all // 64 * 64
        # [h, sketch] inputs: 128 + 3 (or 256 + 3 ...) channels.  As one launch the 3 extra columns cost a whole 64-column
        # tile of matrix work (131 -> 192 columns computed); split: the multiple of 64 on the tile kernel, the rest (<= 4
        # columns) on the narrow kernel (SSC_MRU_SPLIT_DGRAD=0: one launch)
        if _SPLIT_DGRAD and k_real is None and n_off == 0 and 0 < n_all - wide <= 4 and wide >= 64 and n_st - wide <= 4 \
                and dy.shape[-1] % 32 == 0:
            hip.conv_dgrad(View(dy), w, 1, (k - 1) // 2, out, n_off=0, nn=wide, nstore=wide, accumulate=accumulate)
            hip.conv_dgrad(View(dy), w, 1, (k - 1) // 2, out, n_off=wide, nn=n_all - wide, nstore=n_st - wide,
                           accumulate=accumulate, coff=wide)
            return
        hip.conv_dgrad(View(dy), w, 1, (k - 1) // 2, out, n_off=n_off, nn=nn, nstore=nstore, accumulate=accumulate,
                       k_real=k_real)

    # ------------------------------------------------------------------ mru_conv_block_v3
    def _conv_block(self, tag, pre, xin, ht, d, labels, tape):
        """Forward of mru_conv_block_v3 + mean_pool (stride 2).  ht [N,h,w,C_h] plain, xin [N,h,w,4]."""
        B = self.b
        N, h, w, ch = ht.shape
        rec = {'kind': 'conv', 'pre': pre, 'tag': tag, 'xin': xin, 'ht': ht, 'ch': ch, 'd': d}
        na, rec['aux_in'] = self._na_part(tag, pre + '/norm_activation_in', ht, labels)
        full = B.get(tag + '/' + pre + '/full', (N, h, w, ch + 4), zero_on_alloc=True)
        hip.concat_parts(full, [na, dict(x=xin, C=3)])
        mm = B.get(tag + '/' + pre + '/rg_mm', (N, 2, ch))
        rg = self._conv(tag, pre + '/update_gate', View(full), ch, 'rg', epi=2, minmax=mm)
        img = self._conv(tag, pre + '/Conv', View(xin), ch, 'img')
        htp = B.get(tag + '/' + pre + '/ht_plus', ht.shape)
        hip.call('ssc_mru_gate_merge', ht, rg, mm, img, htp, N, h * w, ch)
        hin, rec['aux_m'] = self._na_fwd(tag, pre + '/norm_activation_merge_1', htp, labels)
        h1 = self._conv(tag, pre + '/Conv_1', View(hin), d, 'raw', stats=True)
        h1a, rec['aux_1'] = self._na_fwd(tag, pre + '/Conv_1', h1, labels)
        out = self._conv(tag, pre + '/Conv_2', View(h1a), d, 'raw')
        if ch != d:
            self._conv(tag, pre + '/Conv_3', View(ht), d, 'raw', accumulate_into=out)
        else:
            hip.call('ssc_axpy', out, ht, 1.0, out.numel())
        pooled = B.get(tag + '/' + pre + '/pooled', (N, h // 2, w // 2, d))
        hip.call('ssc_mean_pool2', out, d, pooled, d, N, h, w, d)
        rec.update(full=full, rg=rg, mm=mm, img=img, htp=htp, hin=hin, h1=h1, h1a=h1a, pooled=pooled)
        tape.append(rec)
        return pooled

    def _conv_block_backward(self, rec, labels, need_params=True, acc=False, g_x=None):
        """Consumes the gradient slot of rec['pooled']; adds to the slot of rec['ht'] and, when g_x [N,h,w,4] is given,
        adds the gradient w.r.t. the block's image input to it."""
        B = self.b
        tag, pre, ht, ch, d = rec['tag'], rec['pre'], rec['ht'], rec['ch'], rec['d']
        N, h, w, _ = ht.shape
        gb = lambda name, shape: B.get(tag + '/gb/' + pre + '/' + name, shape)
        g_pool = self._gget(rec['pooled'])
        # mean_pool: every position of a 2x2 block receives g/4
        g_sum = gb('g_sum', (N, h, w, d))
        hip.concat_parts(g_sum, [dict(x=g_pool, upsample=True, ab=B.get('const/quarter/%d' % d, (2 * d,)))])
        # out = Conv_2(h1a) + Conv_3(ht)
        self._conv_param_grads(pre + '/Conv_2', View(rec['h1a']), g_sum, 1, need_params, acc, True)
        g_h1a = gb('g_h1a', rec['h1a'].shape)
        self._conv_dgrad(pre + '/Conv_2', g_sum, g_h1a)
        dh1 = gb('dh1', rec['h1'].shape)
        self._na_bwd(tag, pre + '/Conv_1', rec['h1'], rec['aux_1'], labels, g_h1a, d, dh1, False, need_params, acc)
        self._conv_param_grads(pre + '/Conv_1', View(rec['hin']), dh1, 1, need_params, acc, True)
        g_hin = gb('g_hin', rec['hin'].shape)
        self._conv_dgrad(pre + '/Conv_1', dh1, g_hin)
        g_htp = gb('g_htp', rec['htp'].shape)
        self._na_bwd(tag, pre + '/norm_activation_merge_1', rec['htp'], rec['aux_m'], labels, g_hin, ch, g_htp, False,
                     need_params, acc)
        # ht_plus = ht + r * img
        gr, g_img = gb('gr', rec['rg'].shape), gb('g_img', rec['img'].shape)
        hip.call('ssc_mru_gate_merge_backward', g_htp, rec['rg'], rec['mm'], rec['img'], gr, g_img, N, h * w, ch)
        ws = self._ws()
        d_rg = gb('d_rg', rec['rg'].shape)
        hip.call('ssc_minmax_gate_backward', rec['rg'], rec['mm'], gr, N, h * w, ch, d_rg, ws, ws.numel() * 4)
        self._conv_param_grads(pre + '/Conv', View(rec['xin']), g_img, 1, need_params, acc, True)
        self._conv_param_grads(pre + '/update_gate', View(rec['full']), d_rg, 1, need_params, acc, True)
        g_full = gb('g_full', rec['full'].shape)
        self._conv_dgrad(pre + '/update_gate', d_rg, g_full, nn=ch + 3, nstore=ch + 4)
        # d loss / d ht = g_htp (identity term of ht_plus) + projection / identity shortcut + norm_activation_in path
        g_ht = g_htp
        if ch != d:
            self._conv_param_grads(pre + '/Conv_3', View(ht), g_sum, 1, need_params, acc, True)
            self._conv_dgrad(pre + '/Conv_3', g_sum, g_ht, accumulate=True)
        else:
            hip.call('ssc_axpy', g_ht, g_sum, 1.0, g_ht.numel())
        self._na_bwd(tag, pre + '/norm_activation_in', ht, rec['aux_in'], labels, g_full, ch + 4, g_ht, True,
                     need_params, acc)
        slot = self._gget(ht)
        if slot is None:
            self._gset(ht, g_ht)
        else:
            hip.call('ssc_axpy', slot, g_ht, 1.0, g_ht.numel())
        if g_x is not None:
            self._conv_dgrad(pre + '/Conv', g_img, g_x, nn=3, nstore=4, accumulate=True)
            hip.call('ssc_strided_copy', g_full.view(-1)[ch:], ch + 4, g_x, 4, N * h * w, 3, 1)


class MRUGenerator(_MRUBlocks):
    scope = 'generator'
    cond_norm = True

    def __init__(self, store, bufs, lstm_hybrid=True):
        self._init(store, bufs)
        self.lstm_hybrid = bool(lstm_hybrid)
        self.text_stream = None     # set by the trainer: side stream for the image-independent half of the caption branch
        self.text = TextFusion(store, bufs)
        for d in (64, 128, 256, 512):
            t = bufs.get('const/quarter/%d' % d, (2 * d,), zero_on_alloc=True)
            t[:d].fill_(0.25)

    # ------------------------------------------------------------------ mru_deconv_block_v2
    def _deconv_block(self, tag, pre, z, skip, ht, d, labels, tape):
        """ht [N,h,w,C_h] -> [N,2h,2w,d]."""
        B = self.b
        N, h, w, ch = ht.shape
        H, W = 2 * h, 2 * w
        inp = [dict(x=z, C=3)] + ([dict(x=skip)] if skip is not None else [])
        cs = skip.shape[-1] if skip is not None else 0
        ct = ch + 3 + cs
        full = B.get(tag + '/' + pre + '/full', (N, H, W, _pad4(ct)), zero_on_alloc=True)
        hip.concat_parts(full, [dict(x=ht, upsample=True)] + inp)
        mm_r = B.get(tag + '/' + pre + '/rg_mm', (N, 2, ch))
        rg = self._conv(tag, pre + '/Conv', View(full), ch, 'raw', epi=2, minmax=mm_r)
        mm_z = B.get(tag + '/' + pre + '/zg_mm', (N, 2, d))
        zg = self._conv(tag, pre + '/Conv_1', View(full), d, 'raw', epi=2, minmax=mm_z)
        in2 = B.get(tag + '/' + pre + '/in2', (N, H, W, _pad4(ct)), zero_on_alloc=True)
        hip.concat_parts(in2, [dict(x=ht, upsample=True, gate=(rg, mm_r))] + inp)
        h1 = self._conv(tag, pre + '/Conv_2', View(in2), d, 'raw', stats=True)
        h1a, aux1 = self._na_fwd(tag, pre + '/Conv_2', h1, labels)
        h2 = self._conv(tag, pre + '/Conv_3', View(h1a), d, 'raw', stats=True)
        aux2 = self._cbn(tag, pre + '/Conv_3', h2, labels)
        out = B.get(tag + '/' + pre + '/out', (N, H, W, d))
        pj = auxp = None
        if ch != d:
            pj = self._conv(tag, pre + '/Conv_4', View(ht), d, 'raw', stats=True)        # at low resolution (see module doc)
            auxp = self._cbn(tag, pre + '/Conv_4', pj, labels)
            hip.call('ssc_mru_blend', pj, auxp[0], 1, h2, aux2[0], zg, mm_z, out, N, H, W, d)
        else:
            hip.call('ssc_mru_blend', ht, None, 1, h2, aux2[0], zg, mm_z, out, N, H, W, d)
        tape.append({'kind': 'deconv', 'pre': pre, 'tag': tag, 'z': z, 'skip': skip, 'ht': ht, 'ch': ch, 'd': d, 'cs': cs,
                     'full': full, 'rg': rg, 'mm_r': mm_r, 'zg': zg, 'mm_z': mm_z, 'in2': in2, 'h1': h1, 'aux1': aux1,
                     'h1a': h1a, 'h2': h2, 'aux2': aux2, 'pj': pj, 'auxp': auxp, 'out': out})
        return out

    def _deconv_block_backward(self, rec, labels):
        """Consumes the slot of rec['out']; adds to the slots of rec['ht'] (low resolution) and rec['skip']."""
        B = self.b
        tag, pre, ht, ch, d, cs = rec['tag'], rec['pre'], rec['ht'], rec['ch'], rec['d'], rec['cs']
        N, h, w, _ = ht.shape
        H, W = 2 * h, 2 * w
        P = H * W
        gb = lambda name, shape: B.get(tag + '/gb/' + pre + '/' + name, shape)
        ws = self._ws()
        g_out = self._gget(rec['out'])
        pj, auxp, aux2 = rec['pj'], rec['auxp'], rec['aux2']
        ghp, gh, gz = gb('ghp', g_out.shape), gb('gh', g_out.shape), gb('gz', g_out.shape)
        hip.call('ssc_mru_blend_backward', g_out, pj if pj is not None else ht, auxp[0] if pj is not None else None, 1,
                 rec['h2'], aux2[0], rec['zg'], rec['mm_z'], ghp, gh, gz, N, H, W, d)
        # state path: hp = up(miu(cbn(Conv_4(ht)))) or up(ht)
        g_ht, a_ht = self._gslot(ht)
        if pj is not None:
            g_pja = gb('g_pja', pj.shape)
            hip.call('ssc_pool2', ghp, d, g_pja, d, N, H, W, d, 1.0, 0)
            dpj = gb('dpj', pj.shape)
            self._na_bwd(tag, pre + '/Conv_4', pj, auxp, labels, g_pja, d, dpj, False, True, False)
            self._conv_param_grads(pre + '/Conv_4', View(ht), dpj, 1, True, False, False)
            self._conv_dgrad(pre + '/Conv_4', dpj, g_ht, accumulate=a_ht)
        else:
            hip.call('ssc_pool2', ghp, d, g_ht, ch, N, H, W, d, 1.0, int(a_ht))
        # h_new path
        dh2 = gb('dh2', rec['h2'].shape)
        self._na_bwd(tag, pre + '/Conv_3', rec['h2'], aux2, labels, gh, d, dh2, False, True, False)
        self._conv_param_grads(pre + '/Conv_3', View(rec['h1a']), dh2, 1, True, False, True)
        g_h1a = gb('g_h1a', rec['h1a'].shape)
        self._conv_dgrad(pre + '/Conv_3', dh2, g_h1a)
        dh1 = gb('dh1', rec['h1'].shape)
        self._na_bwd(tag, pre + '/Conv_2', rec['h1'], rec['aux1'], labels, g_h1a, d, dh1, False, True, False)
        self._conv_param_grads(pre + '/Conv_2', View(rec['in2']), dh1, 1, True, False, True)
        ct = ch + 3 + cs
        G = gb('G', rec['in2'].shape)           # d loss / d [ (rg*)up(ht) | z | skip ], all three convs accumulate here
        self._conv_dgrad(pre + '/Conv_2', dh1, G, nn=ct, nstore=_pad4(ct))
        gr = gb('gr', rec['rg'].shape)
        hip.call('ssc_mru_in2_gate_backward', G, G.shape[-1], rec['rg'], rec['mm_r'], ht, gr, N, H, W, ch)
        d_rg = gb('d_rg', rec['rg'].shape)
        hip.call('ssc_minmax_gate_backward', rec['rg'], rec['mm_r'], gr, N, P, ch, d_rg, ws, ws.numel() * 4)
        d_zg = gb('d_zg', rec['zg'].shape)
        hip.call('ssc_minmax_gate_backward', rec['zg'], rec['mm_z'], gz, N, P, d, d_zg, ws, ws.numel() * 4)
        fullv = View(rec['full'])
        self._conv_param_grads(pre + '/Conv', fullv, d_rg, 1, True, False, True)
        self._conv_param_grads(pre + '/Conv_1', fullv, d_zg, 1, True, False, True)
        self._conv_dgrad(pre + '/Conv', d_rg, G, nn=ct, nstore=_pad4(ct), accumulate=True)
        self._conv_dgrad(pre + '/Conv_1', d_zg, G, nn=ct, nstore=_pad4(ct), accumulate=True)
        # split: nearest upsample backward for the state, channel slice for the skip tensor
        hip.call('ssc_pool2', G, G.shape[-1], g_ht, ch, N, H, W, ch, 1.0, 1)
        if rec['skip'] is not None:
            slot, a = self._gslot(rec['skip'])
            hip.call('ssc_strided_copy', G.view(-1)[ch + 3:], G.shape[-1], slot, cs, N * P, cs, int(a))

    # ------------------------------------------------------------------ forward
    def forward(self, sketches, text, labels, noise_vec, tag='g', out=None, out_coff=0):
        """sketches NCHW [N,3,H,W] (device), text int [N,T] (host), labels int32 [N] (device) = class ids,
        noise_vec [N,256] (device).  The tanh image goes to ``out[..., out_coff:out_coff+3]`` (NHWC)."""
        s, B = self.s, self.b
        self._pre_stats.clear()     # entries are keyed by buffer address: none may outlive the pass that made it
        nhwc_in = sketches.dim() == 4 and sketches.shape[3] == 4 and sketches.shape[1] != 3     # hip.sketch_preprocess_u8
        N, H, W = (sketches.shape[0], sketches.shape[1], sketches.shape[2]) if nhwc_in else \
            (sketches.shape[0], sketches.shape[2], sketches.shape[3])
        assert H % 32 == 0 and W % 32 == 0
        labels = labels.to(device=sketches.device, dtype=torch.int32).contiguous()
        if self.lstm_hybrid and self.text_stream is not None and hip.PROFILE is None:
            # the caption's word LSTM does not see the image: start it next to the encoder
            text = self.text.start_words(text, None, tag, self.text_stream)
        tape = []
        if nhwc_in:
            xs = sketches
        else:
            xs = B.get(tag + '/xs', (N, H, W, 4), zero_on_alloc=True)
            hip.nchw_to_nhwc(sketches, xs, 0)
        pyr = [xs]          # mean-pool pyramid == AREA resize for the integer factors (models_collection.py:76-80, 264-267)
        for k in range(1, 5):
            p = B.get(tag + '/pyr%d' % k, (N, H >> k, W >> k, 4))
            hip.call('ssc_mean_pool2', pyr[-1], 4, p, 4, N, H >> (k - 1), W >> (k - 1), 4)
            pyr.append(p)
        h0 = self._conv(tag, 'generator/Conv', View(xs), 8, 'raw', stride=2)
        enc = [h0]
        ht = h0
        aux_last = None
        for (u, ch, d), xin in zip(ENC_UNITS, pyr[1:5]):
            ht = self._conv_block(tag, 'generator/mru_conv_unit_t_%d_layer_0' % u, xin, ht, d, labels, tape)
            if u == 4:      # last_unit (mru.py:651-653)
                pre_last = ht
                ht, aux_last = self._na_fwd(tag, 'generator/mru_conv_unit_last_norm', ht, labels)
            enc.append(ht)
        e5 = enc[-1]
        ctx = {'tag': tag, 'N': N, 'H': H, 'W': W, 'enc': enc, 'xs': xs, 'labels': labels, 'noise_vec': noise_vec,
               'pre_last': pre_last, 'aux_last': aux_last, 'tape': tape}
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
            ht = self._deconv_block(tag, 'generator/mru_deconv_unit_t_%d_layer_0' % u, z, skip, ht, d, labels, tape)
            dec.append(ht)
        if out is None:
            out = B.get(tag + '/gen', (N, H, W, 4), zero_on_alloc=True)
            out_coff = 0
        nstore = 4 if out.shape[3] == 4 and out_coff == 0 else 3
        # the 7x7 64 -> 3 conv on the few-output kernel with its filter through scalar loads: that form wants the input
        # channels of a (tap, output) contiguous, i.e. a transposed copy of the 9408-float filter (one tiny launch)
        w7 = s['generator/Conv_1/weights']
        w7t = hip.transpose_filter(w7, B.get(tag + '/Conv_1/w_nk', (w7.shape[0], w7.shape[1], w7.shape[3], w7.shape[2])))
        hip.conv_forward(View(ht), w7, 1, 0, out, coff=out_coff, nstore=nstore,
                         bias=s['generator/Conv_1/biases'], epi=1, same=True, w_nk=w7t)
        ctx.update(out=out, out_coff=out_coff, feat=feat, dec=dec, noise=noise, noise_pre=pre)
        return ctx

    def output_nchw(self, ctx):
        N, H, W = ctx['N'], ctx['H'], ctx['W']
        o = torch.empty((N, 3, H, W), dtype=torch.float32, device=ctx['out'].device)
        hip.nhwc_to_nchw(ctx['out'], o, ctx['out_coff'])
        return o

    # ------------------------------------------------------------------ backward
    def backward(self, ctx, dpre, on_section=None):
        """dpre [N,H,W,4]: gradient w.r.t. the pre-tanh output of the final 7x7 conv.  Writes every generator gradient
        (and adds the block convs' l2 regularisation, loss and gradient)."""
        s, B = self.s, self.b
        done = on_section if on_section is not None else (lambda name: None)
        tag, N, labels = ctx['tag'], ctx['N'], ctx['labels']
        tape, enc, dec = ctx['tape'], ctx['enc'], ctx['dec']
        self._gdone = {}
        # final conv 7x7 64 -> 3 (+ bias, tanh handled by the caller)
        ht5 = dec[-1]
        self._conv_param_grads('generator/Conv_1', View(ht5), dpre, 1, True, False, False)
        g5, _ = self._gslot(ht5)
        hip.conv_dgrad(View(dpre), s['generator/Conv_1/weights'], 1, 3, g5, k_real=3)
        for rec in reversed(tape[4:]):
            self._deconv_block_backward(rec, labels)
        # noise head
        noise, feat = ctx['noise'], ctx['feat']
        g_noise, g_feat = self._gget(noise), self._gget(feat)
        P = noise.shape[1] * noise.shape[2]
        dpre_fc = B.get(tag + '/gb/noise_dpre', (N, 64 * P))
        hip.call('ssc_miu_permute_bwd', ctx['noise_pre'], g_noise, N, 64, P, dpre_fc)
        hip.matmul_tn(ctx['noise_vec'], dpre_fc, s.grad('generator/fully_connected/weights'))
        hip.call('ssc_group_rowsum', dpre_fc, 64 * P, 1, N, 64 * P, s.grad('generator/fully_connected/biases'), 0)
        done('decoders')
        # caption branch
        e5 = enc[-1]
        if self.lstm_hybrid:
            dy5 = self.text.backward(ctx['tctx'], g_feat)
            if dy5 is None:
                ge5 = B.get(tag + '/gb/ge5', e5.shape)
                hip.fill(ge5, 0.0)
            else:
                ge5 = dy5.view(e5.shape)
            self._gset(e5, ge5)
        else:
            for nm in ('embedding', 'RNN/WLSTM/multi_rnn_cell/cell_0/basic_lstm_cell/kernel',
                       'RNN/WLSTM/multi_rnn_cell/cell_0/basic_lstm_cell/bias',
                       'RNN/ALSTM/multi_rnn_cell/cell_0/basic_lstm_cell/kernel',
                       'RNN/ALSTM/multi_rnn_cell/cell_0/basic_lstm_cell/bias'):
                hip.fill(s.grad('generator/TextLSTM/' + nm), 0.0)
        done('text')
        # encoder: last norm, then the four conv blocks in reverse
        pre_last = ctx['pre_last']
        g_pl = B.get(tag + '/gb/g_pre_last', pre_last.shape)
        self._na_bwd(tag, 'generator/mru_conv_unit_last_norm', pre_last, ctx['aux_last'], labels, self._gget(e5),
                     e5.shape[-1], g_pl, False, True, False)
        self._gset(pre_last, g_pl)
        for rec in reversed(tape[:4]):
            self._conv_block_backward(rec, labels)
        h0 = enc[0]
        self._conv_param_grads('generator/Conv', View(ctx['xs']), self._gget(h0), 2, True, False, False)
        done('encoders')


class MRUDiscriminator(_MRUBlocks):
    """discriminate_mru (models_collection.py:676-786): 7x7 stem + four mru_conv units on the target's mean-pool
    pyramid (the sketch input is ignored by the reference, :690-700), spectral norm on every weight, prelu, no norm;
    1x1 patch head [N,1,12,12] and a class head on the spatial mean.  Interface of Pix2PixDiscriminator."""

    scope = 'discriminator'
    cond_norm = False

    def __init__(self, store, bufs, sn=True):
        self._init(store, bufs)
        if not sn:
            raise NotImplementedError('Config.sn=False for the MRU discriminator')
        self.sn = True
        for d in (4, 128, 256, 512, 768):
            t = bufs.get('const/quarter/%d' % d, (2 * d,), zero_on_alloc=True)
            t[:d].fill_(0.25)
        self.sn_names = ['discriminator/Conv']
        for u, _, _ in DISC_UNITS:
            pre = 'discriminator/mru_conv_unit_t_%d_layer_0' % u
            self.sn_names += [pre + '/' + c for c in ('update_gate', 'Conv', 'Conv_1', 'Conv_2', 'Conv_3')]
        self.sn_names += ['discriminator/Conv_1', 'discriminator/fully_connected']

    # ------------------------------------------------------------------ spectral norm of every weight
    def prepare_sn(self):
        s, B = self.s, self.b
        ws = self._ws()
        sn = {}
        for pre in self.sn_names:
            W = s[pre + '/weights']
            n = W.shape[-1]
            m = W.numel() // n
            st = {'v': B.get('d/sn/' + pre + '/v', (m,)), 'u_new': B.get('d/sn/' + pre + '/u_new', (1, n)),
                  'wbar': B.get('d/sn/' + pre + '/wbar', (m, n)), 'aux': B.get('d/sn/' + pre + '/aux', (4,)),
                  'gwbar': B.get('d/sn/' + pre + '/gwbar', (m, n)), 'n_acc': 0, 'm': m, 'n': n}
            hip.call('ssc_sn_forward_any', W, s[pre + '/u'], m, n, st['v'], st['u_new'], st['wbar'], st['aux'], ws,
                     ws.numel() * 4)
            sn[pre] = st
        return sn

    def finish_sn_backward(self, sn, accumulate=False):
        """d loss / d W from the accumulated d loss / d (W / sigma), plus the block convs' l2 regulariser."""
        s, B = self.s, self.b
        ws = self._ws()
        for pre in self.sn_names:
            st = sn[pre]
            W = s[pre + '/weights']
            gW = s.grad(pre + '/weights')
            if st['n_acc'] == 0:
                if not accumulate:
                    hip.fill(gW, 0.0)
            else:
                hip.call('ssc_sn_backward_any', W, s[pre + '/u'], st['v'], st['u_new'], st['aux'], st['gwbar'], st['m'],
                         st['n'], gW, int(accumulate), B.get('d/sn/' + pre + '/scratch', (st['m'],)), ws,
                         ws.numel() * 4)
            if '/mru_conv_unit_t_' in pre and self.loss_acc is not None:
                hip.call('ssc_l2_reg', W, W.numel(), REG_CONV, self.loss_acc, gW)

    def commit_u(self, sn):
        for pre in self.sn_names:
            self.s[pre + '/u'].copy_(sn[pre]['u_new'])

    # ------------------------------------------------------------------ forward / backward
    def forward(self, xd, sn, tag):
        """xd NHWC [N,H,W,8] = [discrim_inputs(3), discrim_targets(3), 0, 0]; only the targets are used."""
        s, B = self.s, self.b
        N, H, W, _ = xd.shape
        self._sn = sn
        self._pre_stats.clear()
        tape = []
        x0 = B.get(tag + '/x0', (N, H, W, 4), zero_on_alloc=True)
        hip.call('ssc_strided_copy', xd.view(-1)[3:], 8, x0, 4, N * H * W, 3, 0)
        pyr = [x0]
        for k in range(1, 4):
            p = B.get(tag + '/pyr%d' % k, (N, H >> k, W >> k, 4))
            hip.call('ssc_mean_pool2', pyr[-1], 4, p, 4, N, H >> (k - 1), W >> (k - 1), 4)
            pyr.append(p)
        h0raw = self._conv(tag, 'discriminator/Conv', View(x0), 8, 'raw')
        h0, _ = self._na_fwd(tag, 'discriminator/Conv', h0raw, None)
        ht = h0
        for (u, ch, d), xin in zip(DISC_UNITS, pyr):
            ht = self._conv_block(tag, 'discriminator/mru_conv_unit_t_%d_layer_0' % u, xin, ht, d, None, tape)
        pre_last = ht
        img, _ = self._na_fwd(tag, 'discriminator/mru_conv_unit_last_norm', pre_last, None)
        disc = self._conv(tag, 'discriminator/Conv_1', View(img), 1, 'disc', nstore=4)
        P4 = img.shape[1] * img.shape[2]
        C = img.shape[-1]
        mean = B.get(tag + '/img_mean', (N, C))
        hip.call('ssc_act_mean_hw', img, None, ACT_NONE, N, P4, C, mean)
        K = s['discriminator/fully_connected/weights'].shape[1]
        logits = B.get(tag + '/logits', (N, K))
        hip.call('ssc_fc_small_fwd', mean, sn['discriminator/fully_connected']['wbar'],
                 s['discriminator/fully_connected/biases'], N, C, K, logits)
        return {'tag': tag, 'N': N, 'xd': xd, 'x0': x0, 'pyr': pyr, 'h0raw': h0raw, 'h0': h0, 'tape': tape,
                'pre_last': pre_last, 'img': img, 'mean': mean, 'logits': logits, 'disc': disc, 'P4': P4}

    def backward(self, ctx, dl5, dlogits, sn, need_params, need_input, accumulate):
        """dl5 [N,h,w,4] (channel 0 real) = d loss / d patch logits, dlogits [N,K] or None."""
        s, B = self.s, self.b
        tag, N, img, pre_last = ctx['tag'], ctx['N'], ctx['img'], ctx['pre_last']
        self._sn = sn
        self._gdone = {}
        C = img.shape[-1]
        # patch head: 1x1 conv 768 -> 1
        self._conv_param_grads('discriminator/Conv_1', View(img), dl5, 1, need_params, accumulate, False)
        g_img = B.get(tag + '/gb/g_img', img.shape)
        self._conv_dgrad('discriminator/Conv_1', dl5, g_img, k_real=1)
        if dlogits is not None:
            st = sn['discriminator/fully_connected']
            dmean = B.get(tag + '/gb/dmean', (N, C))
            K = dlogits.shape[1]
            gw, gb_, acc = None, None, 0
            if need_params:
                gb_ = s.grad('discriminator/fully_connected/biases')
                gw, acc = st['gwbar'], int(st['n_acc'] > 0)
                st['n_acc'] += 1
            hip.call('ssc_fc_small_bwd', ctx['mean'], st['wbar'], dlogits, N, C, K, dmean, gw, gb_, acc)
            hip.call('ssc_add_row_bcast', g_img, dmean, 1.0 / ctx['P4'], N, ctx['P4'], C)
        g_pl = B.get(tag + '/gb/g_pre_last', pre_last.shape)
        self._na_bwd(tag, 'discriminator/mru_conv_unit_last_norm', pre_last, None, None, g_img, C, g_pl, False,
                     need_params, accumulate)
        self._gset(pre_last, g_pl)
        pyr = ctx['pyr']
        g_pyr = None
        if need_input:
            g_pyr = [B.get(tag + '/gb/g_pyr%d' % k, p.shape) for k, p in enumerate(pyr)]
            for g in g_pyr:
                hip.fill(g, 0.0)
        for k in (3, 2, 1, 0):
            self._conv_block_backward(ctx['tape'][k], None, need_params, accumulate,
                                      g_pyr[k] if need_input else None)
        # stem: h0 = prelu(conv7x7(x0) + b)
        h0raw = ctx['h0raw']
        dh0 = B.get(tag + '/gb/dh0', h0raw.shape)
        self._na_bwd(tag, 'discriminator/Conv', h0raw, None, None, self._gget(ctx['h0']), 8, dh0, False, need_params,
                     accumulate)
        self._conv_param_grads('discriminator/Conv', View(ctx['x0']), dh0, 1, need_params, accumulate, False)
        if not need_input:
            return None
        self._conv_dgrad('discriminator/Conv', dh0, g_pyr[0], nn=3, nstore=4, accumulate=True)
        # back through the mean-pool pyramid: level k-1 += up(level k) / 4
        for k in (3, 2, 1):
            hb = B.get(tag + '/gb/g_pyr_up%d' % k, pyr[k - 1].shape)
            hip.concat_parts(hb, [dict(x=g_pyr[k], upsample=True, ab=B.get('const/quarter/4', (8,)))])
            hip.call('ssc_axpy', g_pyr[k - 1], hb, 1.0, hb.numel())
        return g_pyr[0]


class MRUTower(object):
    """Inference-only tower kept for callers that want the generator without optimizer state."""

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
