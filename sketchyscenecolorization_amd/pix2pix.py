"""Pix2Pix-variant generator and discriminator (models_collection.py:408-538, 789-841)
as explicit forward / hand-written backward passes over the HIP kernels.

Layout: activations NHWC fp32 (3-channel images padded to 4, the 6-channel
discriminator input to 8).  Every conv output is stored RAW (pre-norm); the
batch-statistics norm is folded to a per-channel (a, b) pair by ``bn_stats`` and
applied, together with the lrelu/relu of the *next* block and the skip concat,
inside that block's tile loads (hip.View).  Nothing normalised, activated or
concatenated is ever written to HBM.
"""
import os

import torch

from . import hip
from .hip import _dev_env
from .hip import ACT_LRELU, ACT_NONE, ACT_RELU, View
from .text_fusion import TextFusion


def _rows(t):
    return t if t.dim() == 2 else t.view(-1, t.shape[-1])


_MERGE_DDGRAD = _dev_env('SSC_MERGE_DDGRAD', '1') == '1'
_DBWD_WGRAD_FIRST = _dev_env('SSC_DBWD_WGRAD_FIRST', '1') == '1'   # discriminator backward: filter gradient of layer k in front of its data gradient (round-3 order; the round-4 order -- behind the data gradient and the sums' fold -- measured 0.3 ms slower per iteration on one box, profiles/r05_ab_envsets.txt)
_BNBWD2 = _dev_env('SSC_BNBWD2', '1') == '1'      # norm-backward sums of both halves out of the merged launch's epilogue (A/B)


class Pix2PixGenerator(object):
    """generate_pix2pix: 5 stride-2 convs, caption fusion, noise head, 5 stride-2 transposed convs."""

    def __init__(self, store, bufs, lstm_hybrid=True):
        self.s, self.b = store, bufs
        self.lstm_hybrid = bool(lstm_hybrid)
        # set by the trainer: side stream for the image-independent half of the caption branch, forward / backward.
        # (Backward only when no gradient section ends a graph segment in between: the fork must be joined inside it.)
        self.text_stream = None
        self.text_stream_bwd = None
        self.text = TextFusion(store, bufs)

    def forward(self, sketches, text, noise_vec, tag='g', out=None, out_coff=0):
        """sketches NCHW [N,3,H,W] (device), text int [N,T] (host), noise_vec [N,256] (device).
        Writes tanh output into ``out[..., out_coff:out_coff+3]`` (NHWC) and returns the context."""
        s, B = self.s, self.b
        nhwc_in = sketches.dim() == 4 and sketches.shape[3] == 4 and sketches.shape[1] != 3     # hip.sketch_preprocess_u8
        N, H, W = (sketches.shape[0], sketches.shape[1], sketches.shape[2]) if nhwc_in else \
            (sketches.shape[0], sketches.shape[2], sketches.shape[3])
        chans = [None, 64, 128, 256, 512, 512]
        tstream = self.text_stream if hip.PROFILE is None else None     # per-kernel timing runs everything in line
        hh = H >> 5             # encoder_5's output (five stride-2 layers)
        ww = hh * W // H
        P = hh * ww
        cd = chans[5] // 8

        def noise_head():
            # fully_connected(256 -> 64*P) + miu_relu, reshaped NCHW->NHWC
            pre = B.get(tag + '/noise_pre', (N, cd * P))
            hip.matmul(noise_vec, s['generator/fully_connected/weights'], pre, bias=s['generator/fully_connected/biases'])
            noise = B.get(tag + '/noise', (N, hh, ww, cd))
            hip.call('ssc_miu_permute_fwd', pre, N, cd, P, noise)
            return pre, noise

        early = None
        if self.lstm_hybrid and tstream is not None:
            # the caption's word LSTM does not see the image: start it next to the encoder convolutions
            text = self.text.start_words(text, chans[5], tag, tstream)
            if isinstance(text, dict) and text.get('words_stream') is tstream:
                # ... and neither does the noise head: behind the word half on its stream instead of between the recurrence
                # and decoder_5 on the main chain (text.forward joins that stream in front of the recurrence)
                with torch.cuda.stream(tstream):
                    early = noise_head()
        if nhwc_in:
            xs = sketches
        else:
            xs = B.get(tag + '/xs', (N, H, W, 4), zero_on_alloc=True)
            hip.nchw_to_nhwc(sketches, xs, 0)
        e, ab, st = [None] * 6, [None] * 6, [None] * 6
        h = H
        for k in range(1, 6):
            h //= 2
            e[k] = B.get(tag + '/e%d' % k, (N, h, h * W // H, chans[k]))
            if k == 1:
                hip.conv_forward(View(xs), s['generator/encoder_1/conv/filter'], 2, 1, e[1])
            else:
                ab[k] = B.get(tag + '/ab_e%d' % k, (2 * chans[k],))
                st[k] = B.get(tag + '/st_e%d' % k, (2 * chans[k],))
                # conv + batch statistics of its output in one launch (the sums come out of the conv epilogue)
                hip.conv_forward(View(e[k - 1], None, ab[k - 1], ACT_LRELU), s['generator/encoder_%d/conv/filter' % k],
                                 2, 1, e[k], bn=(s['generator/encoder_%d/scale' % k], s['generator/encoder_%d/offset' % k],
                                                 ab[k], st[k]))
        hip.mark(tag + '/encoders: last')
        ctx = {'tag': tag, 'N': N, 'H': H, 'W': W, 'xs': xs, 'e': e, 'ab': ab, 'st': st, 'noise_vec': noise_vec}
        assert (hh, ww) == (e[5].shape[1], e[5].shape[2])
        if self.lstm_hybrid:
            feat, tctx = self.text.forward(e[5], ab[5], text, tag)
            ctx['tctx'] = tctx
            v5 = lambda noise: View(feat, noise, None, ACT_RELU, None)
        else:
            feat = e[5]
            v5 = lambda noise: View(feat, noise, ab[5], ACT_RELU, None)
        ctx['feat'] = feat
        pre, noise = early if early is not None else noise_head()
        ctx['noise_pre'], ctx['noise'] = pre, noise
        # decoders
        hip.mark(tag + '/caption fusion: last')
        d, abd, std = [None] * 6, [None] * 6, [None] * 6
        dch = [None, 3, 64, 128, 256, 512]
        views = {}
        for k in (5, 4, 3, 2):
            if k == 5:
                v = v5(noise)
            else:
                v = View(d[k + 1], e[k], abd[k + 1], ACT_RELU, ab[k])
            views[k] = v
            d[k] = B.get(tag + '/d%d' % k, (N, 2 * v.H, 2 * v.W, dch[k]))
            abd[k] = B.get(tag + '/ab_d%d' % k, (2 * dch[k],))
            std[k] = B.get(tag + '/st_d%d' % k, (2 * dch[k],))
            hip.deconv_forward(v, s['generator/decoder_%d/deconv/filter' % k], d[k],
                               bn=(s['generator/decoder_%d/scale' % k], s['generator/decoder_%d/offset' % k], abd[k], std[k]))
        v1 = View(d[2], e[1], abd[2], ACT_RELU, None)
        views[1] = v1
        if out is None:
            out = B.get(tag + '/gen', (N, H, W, 4))
            out_coff = 0
        nstore = 4 if out.shape[3] == 4 and out_coff == 0 else 3
        hip.deconv_forward(v1, s['generator/decoder_1/deconv/filter'], out, coff=out_coff, nstore=nstore, epi=1)
        ctx.update(d=d, abd=abd, std=std, views=views, out=out, out_coff=out_coff)
        return ctx

    def output_nchw(self, ctx):
        N, H, W = ctx['N'], ctx['H'], ctx['W']
        o = torch.empty((N, 3, H, W), dtype=torch.float32, device=ctx['out'].device)
        hip.nhwc_to_nchw(ctx['out'], o, ctx['out_coff'])
        return o

    bn_stream = None        # helper stream of backward (set by the trainer): see _fork

    def _fork(self, fn):
        """Issue fn's launches on the helper stream, ordered after everything issued so far on the current stream.
        Returns whether a fork happened (then _join() before the next dependent or full-size launch)."""
        st = self.bn_stream if hip.PROFILE is None else None
        if st is None:
            fn()
            return False
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            fn()
        return True

    def _join(self):
        torch.cuda.current_stream().wait_stream(self.bn_stream)

    def _sums(self, tag, name, x, ab, stats, sources=1):
        x2d = _rows(x)
        buf = self.b.get(tag + '/gb/bnsums_' + name, (hip.BnBwdSums.rows_needed(x2d.shape[0], sources), 2 * x2d.shape[1]))
        return hip.BnBwdSums(x2d, ab, stats, buf)

    def backward(self, ctx, dpre, on_section=None, side_stream=None):
        """dpre [N,H,W,4]: gradient w.r.t. the pre-tanh output.  Writes every generator gradient.
        ``on_section(name)`` is called when a contiguous block of the flat gradient buffer is final
        ('decoders' = noise head + decoders, 'text', 'encoder_5', 'encoders' = encoder_1..4) so the caller can start its all-reduce.
        With ``side_stream`` the decoder filter gradients (off the critical path: nothing downstream reads
        them) are held back and launched on that stream next to the caption branch's BPTT, whose small
        recurrent GEMMs leave most CUs idle."""
        s, B = self.s, self.b
        held = []
        done = on_section if on_section is not None else (lambda name: None)
        tag, N = ctx['tag'], ctx['N']
        e, ab, st, d, abd, std, views = ctx['e'], ctx['ab'], ctx['st'], ctx['d'], ctx['abd'], ctx['std'], ctx['views']
        gcur = dpre
        g_skip = [None] * 6
        sums_e = [None] * 6
        g_feat = g_noise = None
        hold = side_stream is not None and self.lstm_hybrid
        for k in (1, 2, 3, 4, 5):
            f = s['generator/decoder_%d/deconv/filter' % k]
            v = views[k]
            dyv = View(gcur)
            wg = (lambda v=v, dyv=dyv, k=k:
                  hip.deconv_wgrad(v, dyv, s.grad('generator/decoder_%d/deconv/filter' % k)))
            if hold:
                held.append(wg)
            g0 = B.get(tag + '/gb/d%d_in0' % k, (N, v.H, v.W, v.C0))
            g1 = B.get(tag + '/gb/d%d_in1' % k, (N, v.H, v.W, v.C1))
            # the two per-channel sums of each norm's backward are taken by the epilogues of the data-gradient launches
            # that produce its incoming gradients (hip.BnBwdSums): decoder_{k+1}'s norm from g0, encoder_k's from g1 + gin
            merged = _MERGE_DDGRAD and 2 <= k <= 4
            sums_d = self._sums(tag, 'd%d' % (k + 1), d[k + 1], abd[k + 1], std[k + 1]) if k < 5 else None
            if 2 <= k <= 4:
                sums_e[k] = self._sums(tag, 'e%d' % k, e[k], ab[k], st[k], sources=2)
            if merged:
                # decoder_k reads concat[decoder_{k+1}, encoder_k]: the gradients of the two halves are the SAME gather of dy
                # against two column ranges of the filter.  One launch over both ranges (twice the tiles: 288 instead of
                # 2 x 144 for decoder_4 -- half-empty launches otherwise) into one buffer, the halves read back through
                # strided row views; 17.77 -> 17.65 ms per step.  The norm-backward sums of BOTH sites come out of its
                # epilogue: a column tile lies in one half, so each workgroup takes the tables of its own tensor.
                g01 = B.get(tag + '/gb/d%d_in01' % k, (N, v.H, v.W, v.C0 + v.C1))
                if _BNBWD2:
                    hip.deconv_dgrad(dyv, f, g01, n_off=0, nn=v.C0 + v.C1,
                                     bnbwd=[sums_d.take(ACT_RELU), sums_e[k].take(ACT_RELU)])
                else:
                    hip.deconv_dgrad(dyv, f, g01, n_off=0, nn=v.C0 + v.C1)
                    for sm in (sums_d, sums_e[k]):
                        sm.sources += 1
                        sm.missed += 1
                r01 = g01.view(-1, v.C0 + v.C1)
                g0, g1 = r01[:, :v.C0], r01[:, v.C0:]
            else:
                hip.deconv_dgrad(dyv, f, g0, n_off=0, nn=v.C0, bnbwd=(sums_d.take(ACT_RELU) if sums_d else None))

            def rest(wg=wg, dyv=dyv, f=f, g1=g1, v=v, k=k, merged=merged):
                if not merged:
                    hip.deconv_dgrad(dyv, f, g1, n_off=v.C0, nn=v.C1, bnbwd=(sums_e[k].take(ACT_RELU) if sums_e[k] else None))
                if not hold:
                    wg()
            forked = self._fork(rest)
            if k < 5:
                g_skip[k] = g1          # through relu to encoder_k's output
                src = d[k + 1]
                dx = B.get(tag + '/gb/dd%d' % (k + 1), src.shape)
                hip.bn_act_backward(_rows(src), abd[k + 1], std[k + 1], _rows(g0), ACT_RELU, _rows(dx),
                                    dscale=s.grad('generator/decoder_%d/scale' % (k + 1)),
                                    doffset=s.grad('generator/decoder_%d/offset' % (k + 1)), pre=sums_d)
                gcur = dx
            else:
                g_feat, g_noise = g0, g1
            if forked:
                self._join()
        # noise head
        P = ctx['noise'].shape[1] * ctx['noise'].shape[2]
        cd = ctx['noise'].shape[3]
        dpre_fc = B.get(tag + '/gb/noise_dpre', (N, cd * P))
        hip.call('ssc_miu_permute_bwd', ctx['noise_pre'], g_noise, N, cd, P, dpre_fc)
        hip.matmul_tn(ctx['noise_vec'], dpre_fc, s.grad('generator/fully_connected/weights'))
        hip.call('ssc_group_rowsum', dpre_fc, cd * P, 1, N, cd * P, s.grad('generator/fully_connected/biases'), 0)
        hip.mark(tag + '/bwd decoders dgrad: last')
        if held:
            main = torch.cuda.current_stream()
            side_stream.wait_stream(main)
            with torch.cuda.stream(side_stream):
                for wg in held:
                    wg()
        else:
            done('decoders')
        # caption branch -> gradient w.r.t. normalised encoder_5 output
        de5 = B.get(tag + '/gb/de5', e[5].shape)
        text_pending = False
        if self.lstm_hybrid:
            tstream = self.text_stream_bwd if hip.PROFILE is None else None
            dy5 = self.text.backward(ctx['tctx'], g_feat, side_stream=tstream)
            hip.mark(tag + '/bwd caption branch: last (main stream)')
            if held:
                main.wait_stream(side_stream)
                hip.mark(tag + '/bwd held decoder wgrads joined')
                done('decoders')
            if tstream is None:
                done('text')
            else:
                text_pending = True     # its word-branch gradients are still being computed next to the encoder backward
            if dy5 is None:
                hip.fill(de5, 0.0)
                hip.fill(s.grad('generator/encoder_5/scale'), 0.0)
                hip.fill(s.grad('generator/encoder_5/offset'), 0.0)
            else:
                hip.bn_act_backward(_rows(e[5]), ab[5], st[5], dy5, ACT_NONE, _rows(de5),
                                    dscale=s.grad('generator/encoder_5/scale'),
                                    doffset=s.grad('generator/encoder_5/offset'))
        else:
            for nm in ('embedding', 'RNN/WLSTM/multi_rnn_cell/cell_0/basic_lstm_cell/kernel',
                       'RNN/WLSTM/multi_rnn_cell/cell_0/basic_lstm_cell/bias',
                       'RNN/ALSTM/multi_rnn_cell/cell_0/basic_lstm_cell/kernel',
                       'RNN/ALSTM/multi_rnn_cell/cell_0/basic_lstm_cell/bias'):
                hip.fill(s.grad('generator/TextLSTM/' + nm), 0.0)
            done('text')
            hip.bn_act_backward(_rows(e[5]), ab[5], st[5], _rows(g_feat), ACT_RELU, _rows(de5),
                                dscale=s.grad('generator/encoder_5/scale'),
                                doffset=s.grad('generator/encoder_5/offset'))
        gcur = de5
        for k in (5, 4, 3, 2):
            w = s['generator/encoder_%d/conv/filter' % k]
            xin = View(e[k - 1], None, ab[k - 1], ACT_LRELU)
            dyv = View(gcur)
            gin = B.get(tag + '/gb/e%d_in' % k, e[k - 1].shape)
            # (a sums object that already missed a source falls back to the separate pass anyway: do not make this epilogue
            # re-read e[k-1] for rows nobody will use)
            live = sums_e[k - 1] is not None and not sums_e[k - 1].missed
            hip.conv_dgrad(dyv, w, 2, 1, gin, bnbwd=(sums_e[k - 1].take(ACT_LRELU) if live else None))
            dx = B.get(tag + '/gb/de%d' % (k - 1), e[k - 1].shape)
            forked = self._fork(lambda xin=xin, dyv=dyv, k=k:
                                hip.conv_wgrad(xin, dyv, s.grad('generator/encoder_%d/conv/filter' % k), 2, 1))
            if k - 1 >= 2:
                hip.bn_act_backward(_rows(e[k - 1]), ab[k - 1], st[k - 1], _rows(gin), ACT_LRELU, _rows(dx),
                                    g2=_rows(g_skip[k - 1]), act2=ACT_RELU,
                                    dscale=s.grad('generator/encoder_%d/scale' % (k - 1)),
                                    doffset=s.grad('generator/encoder_%d/offset' % (k - 1)), pre=sums_e[k - 1])
            else:
                hip.bn_act_backward(_rows(e[1]), None, None, _rows(gin), ACT_LRELU, _rows(dx),
                                    g2=_rows(g_skip[1]), act2=ACT_RELU)
            if forked:
                self._join()
            gcur = dx
            if k == 5 and not text_pending:
                done('encoder_5')       # its filter, scale and offset gradients are final (trainer._sections)
        hip.conv_wgrad(View(ctx['xs']), View(gcur), s.grad('generator/encoder_1/conv/filter'), 2, 1)
        hip.mark(tag + '/bwd encoders: last')
        if text_pending:
            self.text.join_backward()
            done('text')
            done('encoder_5')
        done('encoders')


_HEAD1_FUSED = _dev_env('SSC_HEAD1_FUSED', '0') == '1'


class Pix2PixDiscriminator(object):
    """discriminate_pix2pix: 70x70-style PatchGAN + spectral-normed auxiliary classifier."""

    def __init__(self, store, bufs, sn=True):
        self.s, self.b = store, bufs
        self.sn = bool(sn)
        self.chans = [8, 64, 128, 256, 512, 1]
        self.strides = [None, 2, 2, 2, 1, 1]

    def prepare_sn(self, tag='d/sn', u=None):
        """One power iteration on fully_connected/weights (sn.py), shared by every call of this step.
        tag / u: buffers and starting vector of a pass that runs beside another step's (the discriminator's real pass run
        ahead inside the generator step starts from the u that step is about to assign: trainer._d_real_pass)."""
        s, B = self.s, self.b
        W = s['discriminator/fully_connected/weights']
        m, n = W.shape
        if not self.sn:
            return {'wbar': W}
        sn = {'v': B.get(tag + '/v', (m,)), 'u_new': B.get(tag + '/u_new', (1, n)), 'wbar': B.get(tag + '/wbar', (m, n)),
              'aux': B.get(tag + '/aux', (4,)), 'gwbar': B.get(tag + '/gwbar', (m, n)), 'n_acc': 0,
              'scratch': B.get(tag + '/scratch', (m,)), 'u': (u if u is not None else s['discriminator/fully_connected/u'])}
        hip.call('ssc_sn_forward', W, sn['u'], m, n, sn['v'], sn['u_new'], sn['wbar'], sn['aux'])
        return sn

    def forward(self, xd, sn, tag):
        """xd NHWC [N,H,W,8] = [discrim_inputs(3), discrim_targets(3), 0, 0]."""
        s, B = self.s, self.b
        N, H, W, _ = xd.shape
        l, ab, st = [None] * 6, [None] * 6, [None] * 6
        l[0] = xd
        h, w = H, W
        for k in range(1, 6):
            if self.strides[k] == 2:
                h, w = h // 2, w // 2
            else:
                h, w = h - 1, w - 1
            co = self.chans[k]
            l[k] = B.get(tag + '/l%d' % k, (N, h, w, 4 if k == 5 else co))
            if k == 1:
                v = View(xd)
            elif k == 2:
                v = View(l[1], None, None, ACT_LRELU)
            else:
                v = View(l[k - 1], None, ab[k - 1], ACT_LRELU)
            bn = None
            if 2 <= k <= 4:
                ab[k] = B.get(tag + '/ab%d' % k, (2 * co,))
                st[k] = B.get(tag + '/st%d' % k, (2 * co,))
                bn = (s['discriminator/layer_%d/scale' % k], s['discriminator/layer_%d/offset' % k], ab[k], st[k])
            hip.conv_forward(v, s['discriminator/layer_%d/conv/filter' % k], self.strides[k], 1, l[k],
                             nstore=(4 if k == 5 else None), bn=bn)
        P4 = l[4].shape[1] * l[4].shape[2]
        img = B.get(tag + '/img', (N, 512))
        hip.call('ssc_act_mean_hw', l[4], ab[4], ACT_LRELU, N, P4, 512, img)
        K = s['discriminator/fully_connected/weights'].shape[1]
        logits = B.get(tag + '/logits', (N, K))
        hip.call('ssc_fc_small_fwd', img, sn['wbar'], s['discriminator/fully_connected/biases'], N, 512, K, logits)
        return {'tag': tag, 'N': N, 'l': l, 'ab': ab, 'st': st, 'img': img, 'logits': logits, 'disc': l[5], 'P4': P4}

    def backward(self, ctx, dl5, dlogits, sn, need_params, need_input, accumulate, after_layer=None, stop_after=None,
                 resume=None):
        """dl5 [N,h5,w5,4] (channel 0 real), dlogits [N,K] or None.
        need_params: write (accumulate=False) or add (True) the filter/norm gradients.
        need_input: return d loss / d discrim_targets as NHWC [N,H,W,4].
        after_layer(k): called when layer k's filter / scale / offset gradients have been launched (the trainer starts the
        all-reduce of a finished section of the flat gradient buffer there).
        stop_after=k: return after layer k's part (its norm backward, filter gradient and the data gradient into layer k-1)
        with the state to go on from; resume=that state: the rest of the pass.  The trainer runs the two passes of a
        discriminator step side by side in two such halves when the gradient of layers 4, 5 and the class head goes to the
        all-reduce between them (more than one tower)."""
        s, B = self.s, self.b
        tag, N, l, ab, st = ctx['tag'], ctx['N'], ctx['l'], ctx['ab'], ctx['st']
        gname = lambda k, what: s.grad('discriminator/layer_%d/%s' % (k, what))
        if resume is not None:      # resume['gcur']: gradient w.r.t. the raw output of the first remaining layer (already applied)
            return self._backward_layers(ctx, resume['layers'], resume['gcur'], resume['sums'], None, None, need_params,
                                         need_input, accumulate, after_layer, None)
        # layer 5 (Cout = 1)
        x5 = View(l[4], None, ab[4], ACT_LRELU)
        dy5 = View(dl5)
        if need_params:
            hip.conv_wgrad(x5, dy5, gname(5, 'conv/filter'), 1, 1, accumulate=accumulate)
        rowb = None
        if dlogits is not None:
            dimg = B.get(tag + '/gb/dimg', (N, 512))
            K = dlogits.shape[1]
            gw, gb_, acc = None, None, 0
            if need_params:
                gb_ = s.grad('discriminator/fully_connected/biases')
                if self.sn:     # gradient w.r.t. W_bar, summed over the calls of this step
                    gw, acc = sn['gwbar'], int(sn['n_acc'] > 0)
                    sn['n_acc'] += 1
                else:
                    gw, acc = s.grad('discriminator/fully_connected/weights'), int(accumulate)
            hip.call('ssc_fc_small_bwd', ctx['img'], sn['wbar'], dlogits, N, 512, K, dimg, gw, gb_, acc)
            # the class head reads the spatial mean of layer 4: its gradient dimg / P4 is added to g4 by the norm backward
            # below while it reads g4 (no pass of its own over the tensor)
            rowb = (dimg, 1.0 / ctx['P4'], ctx['P4'])
        return self._backward_layers(ctx, (4, 3, 2, 1), None, None, dy5, rowb, need_params, need_input, accumulate, after_layer,
                                     stop_after)

    def _norm_backward(self, ctx, k, gcur, sums, need_params, accumulate, rowb=None):
        """The norm + lrelu backward of layer k's output (k = 1: lrelu only).  Returns the dx buffer."""
        s, B = self.s, self.b
        tag, l, ab, st = ctx['tag'], ctx['l'], ctx['ab'], ctx['st']
        gname = lambda kk, what: s.grad('discriminator/layer_%d/%s' % (kk, what))
        dx = B.get(tag + '/gb/dl%d' % k, l[k].shape)
        if k == 1:
            hip.bn_act_backward(_rows(l[1]), None, None, _rows(gcur), ACT_LRELU, _rows(dx))
            return dx
        ds = do = None
        if need_params and accumulate:
            tmp_s = B.get(tag + '/gb/tmp_scale%d' % k, (2, self.chans[k]))
            ds, do = tmp_s[0], tmp_s[1]
        elif need_params:
            ds, do = gname(k, 'scale'), gname(k, 'offset')
        hip.bn_act_backward(_rows(l[k]), ab[k], st[k], _rows(gcur), ACT_LRELU, _rows(dx), dscale=ds, doffset=do, pre=sums,
                            rowb=rowb)
        if need_params and accumulate:
            hip.call('ssc_axpy', gname(k, 'scale'), ds, 1.0, self.chans[k])
            hip.call('ssc_axpy', gname(k, 'offset'), do, 1.0, self.chans[k])
        return dx

    def _backward_layers(self, ctx, layers, dx, sums, dy5, rowb, need_params, need_input, accumulate, after_layer, stop_after):
        """Layers ``layers`` of the backward pass.  dx: gradient w.r.t. the RAW output of layers[0], or None for layer 4 (layer 5's
        data gradient and layer 4's norm backward are taken here).
        Order per layer k: filter gradient of layer k, data gradient into layer k-1 (its epilogue takes the sums of layer k-1's
        norm backward), that norm backward.  (Round 4 put the filter gradient behind the data gradient and had it host the
        norm backward's streaming pass: 0.4 ms slower per iteration on one box, profiles/r05_ab_envsets.txt.)"""
        s, B = self.s, self.b
        tag, N, l, ab, st = ctx['tag'], ctx['N'], ctx['l'], ctx['ab'], ctx['st']
        gname = lambda k, what: s.grad('discriminator/layer_%d/%s' % (k, what))
        dgen = None
        if dx is None:
            assert layers[0] == 4
            dx = B.get(tag + '/gb/dl4', l[4].shape)
            ds = do = None
            if need_params and accumulate:
                tmp_s = B.get(tag + '/gb/tmp_scale4', (2, self.chans[4]))
                ds, do = tmp_s[0], tmp_s[1]
            elif need_params:
                ds, do = gname(4, 'scale'), gname(4, 'offset')
            # d loss / d act(norm(l4)) = layer 5's data gradient + the class head's term.  SSC_HEAD1_FUSED=1 recomputes
            # it inside the two passes of the norm backward instead of storing it: 44 vs 55 us alone, but 0.1 ms
            # SLOWER per iteration in the replayed step (its vector-ALU work competes with the matrix kernels of the
            # chains running beside it; the separate launches are memory traffic those leave idle) -- off by default
            fused = _HEAD1_FUSED and hip.head1_dgrad_bn_backward(dy5, s['discriminator/layer_5/conv/filter'], 1, l[4], ab[4], st[4],
                                                                 ACT_LRELU, dx, dscale=ds, doffset=do,
                                                                 rowb=(rowb[:2] if rowb else None))
            if not fused:
                g4 = B.get(tag + '/gb/g4', l[4].shape)
                hip.conv_dgrad(dy5, s['discriminator/layer_5/conv/filter'], 1, 1, g4, k_real=1)
                hip.bn_act_backward(_rows(l[4]), ab[4], st[4], _rows(g4), ACT_LRELU, _rows(dx), dscale=ds, doffset=do, rowb=rowb)
            if need_params and accumulate:
                hip.call('ssc_axpy', gname(4, 'scale'), ds, 1.0, self.chans[4])
                hip.call('ssc_axpy', gname(4, 'offset'), do, 1.0, self.chans[4])
        for k in layers:
            if k == 1:
                xin = View(l[0])
            elif k == 2:
                xin = View(l[1], None, None, ACT_LRELU)
            else:
                xin = View(l[k - 1], None, ab[k - 1], ACT_LRELU)
            dyv = View(dx)
            w = s['discriminator/layer_%d/conv/filter' % k]
            dx_next = None
            if need_params and _DBWD_WGRAD_FIRST:
                # round-3 order: the filter gradient of layer k in front of its data gradient (the chain's next full-size launch
                # then follows the small launches of the norm backward directly)
                hip.conv_wgrad(xin, dyv, gname(k, 'conv/filter'), self.strides[k], 1, accumulate=accumulate)
                if after_layer is not None:
                    after_layer(k)
            if k > 1:
                gin = B.get(tag + '/gb/g%d' % (k - 1), l[k - 1].shape)
                sums = None
                if k - 1 >= 2:
                    x2d = _rows(l[k - 1])
                    sums = hip.BnBwdSums(x2d, ab[k - 1], st[k - 1],
                                         B.get(tag + '/gb/bnsums%d' % (k - 1),
                                               (hip.BnBwdSums.rows_needed(x2d.shape[0]), 2 * x2d.shape[1])))
                hip.conv_dgrad(dyv, w, self.strides[k], 1, gin, bnbwd=(sums.take(ACT_LRELU) if sums else None))
                dx_next = self._norm_backward(ctx, k - 1, gin, sums, need_params, accumulate)
            elif need_input:
                dgen = B.get(tag + '/gb/dgen', (N, l[0].shape[1], l[0].shape[2], 4))
                hip.conv_dgrad(dyv, w, 2, 1, dgen, n_off=3, nn=3, nstore=4)
            if need_params and not _DBWD_WGRAD_FIRST:
                hip.conv_wgrad(xin, dyv, gname(k, 'conv/filter'), self.strides[k], 1, accumulate=accumulate)
                if after_layer is not None:
                    after_layer(k)
            dx = dx_next
            if stop_after == k and k > 1:
                return {'layers': tuple(j for j in layers if j < k), 'gcur': dx, 'sums': None}
        return dgen

    def finish_sn_backward(self, sn, accumulate=False):
        """d loss / d W from the accumulated d loss / d W_bar, through sigma and the power iteration."""
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
        hip.call('ssc_sn_backward', W, sn['u'], sn['v'], sn['u_new'], sn['aux'], sn['gwbar'], m, n, gW, int(accumulate),
                 sn['scratch'])
