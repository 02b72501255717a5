"""Caption branch: encode_feat_with_text (models_collection.py:150-248) on HIP kernels.

The reference unrolls, per sample, 15 ``tf.cond`` steps of a word LSTM and a
per-position ("1x1 convolutional") multimodal LSTM whose input is the tiled
concat [l2n(visual), w_emb, l2n(h_w)].  Here the whole batch advances together
and the ALSTM matmul is split by kernel row block (only the summation order
changes):

    gates = visual @ Ka[0:C]            (once, all positions)      + bias
          + emb_t  @ Ka[C:2C]           (one row per sample, all steps in one GEMM)
          + lang_t @ Ka[2C:3C]          (idem)
          + h_a    @ Ka[3C:4C]          (the only per-step GEMM)

so the tile/concat tensor is never materialised.  Steps whose token is 0 are
skipped per sample by a mask inside the pointwise kernel (tf.cond, :235); steps
where every sample is padded are not launched at all.  Forward keeps what the
hand-written BPTT needs.
"""
import numpy as np
import torch

from . import hip
from .hip import _dev_env

import os

CELL = '/RNN/%s/multi_rnn_cell/cell_0/basic_lstm_cell/'
FUSED_STEP = _dev_env('SSC_LSTM_FUSED', '1') != '0'     # recurrent GEMM + gate math in one launch per step
# ... for cells of fewer rows than this.  The one-launch step has no operand reuse through LDS (64 x 32 / 64 x 64 tiles straight
# from L2: what makes it the faster form when a step is a handful of tiles per CU) and its gate math sits behind the K loop of
# every workgroup; with many rows the conv kernel's 128 x 128 tiles + the gate kernel are ahead.  us per step alone, one launch
# -> two (scripts/lstm_step_probe.py): C = 512: 576 rows 23.4 -> 26.1, 1152 rows 39.2 -> 33.4, 2304 rows 71.6 -> 56.6;
# C = 1024: 576 rows 55.0 -> 52.3, 1152 rows 93 -> 88, 2304 rows (the Background module at batch 4) 194 -> 138.  In the
# workloads (scripts/lstm_unfused_ab.sh, threshold 1024 vs never, same box, interleaved): Background 768^2 forward 12.96 / 12.99
# -> 12.59 / 12.65 ms; the batch-32 train steps, whose 1152-row cells run beside other chains, do not move (Pix2Pix 12.94 /
# 12.83 / 12.80 vs 12.93 / 12.92 / 12.79 ms, MRU 185.0 / 184.4 vs 185.2 / 185.4): the threshold sits above them.
UNFUSED_ROWS = int(_dev_env('SSC_LSTM_UNFUSED_ROWS', '2048'))
PINNED_PREP = _dev_env('SSC_PINNED_PREP', '1') != '0'       # caption tokens to the device through pinned memory (A/B: 0)


class TextFusion(object):
    def __init__(self, store, bufs, scope='generator/TextLSTM'):
        """scope: 'generator/TextLSTM' (FG, models_collection.py:159) or 'generator/mLSTM_G' (BG,
        bg_colorization_main.py:117-214 -- the same cell pair with C=1024 on the 24x24 bottleneck)."""
        self.s = store
        self.b = bufs
        self.emb_name = scope + '/embedding'
        self.pfx_w = scope + CELL % 'WLSTM'
        self.pfx_a = scope + CELL % 'ALSTM'

    def prepare(self, text, tag='g'):
        """Host side of the caption branch: which steps run at all, time-major token ids and the per-sample
        skip mask, uploaded to (static) device buffers.  Kept apart from ``forward`` so that a captured
        hipGraph of the step contains no host->device copy."""
        B = self.b
        text = np.asarray(text.cpu() if isinstance(text, torch.Tensor) else text).astype(np.int32)
        N = text.shape[0]
        text = text.reshape(N, -1)
        steps = [t for t in range(text.shape[1]) if (text[:, t] != 0).any()]
        S = len(steps)
        prep = {'S': S, 'N': N}
        if S > 0:
            tok_np = np.ascontiguousarray(text[:, steps].T).reshape(-1)        # time-major [S*N]
            # through pinned memory, not waited for: a copy from pageable memory returns when it has HAPPENED, i.e. behind
            # everything queued on the stream -- the caller of a training step would wait here for the previous step to end
            tok = B.get(tag + '/tf/tok', (S * N,), torch.int32)
            pin = (lambda t: t.pin_memory()) if PINNED_PREP else (lambda t: t)
            tok.copy_(pin(torch.from_numpy(tok_np)), non_blocking=PINNED_PREP)
            mask = B.get(tag + '/tf/mask', (S, N), torch.int32)
            mask.copy_(pin(torch.from_numpy((tok_np != 0).astype(np.int32)).view(S, N)), non_blocking=PINNED_PREP)
            prep.update(tok=tok, mask=mask)
        return prep

    def _words_forward(self, prep, C, tag):
        """The part of the branch that does not see the image: embedding, word LSTM over the S live steps, and the two
        word-dependent terms of the multimodal gates for every step (one GEMM each)."""
        s, B = self.s, self.b
        S, N = prep['S'], prep['N']
        tok, mask = prep['tok'], prep['mask']
        E = s[self.emb_name]
        Kw, bw = s[self.pfx_w + 'kernel'], s[self.pfx_w + 'bias']
        Ka = s[self.pfx_a + 'kernel']
        G4 = 4 * C
        emb = B.get(tag + '/tf/emb', (S * N, C))
        hip.call('ssc_embedding_gather', E, tok, S * N, C, emb)
        EW = B.get(tag + '/tf/EW', (S * N, G4))
        hip.matmul(emb, Kw[0:C], EW, bias=bw)
        # ---- word LSTM (state [c,h], batch rows) ----
        # (state 0 of a cell is zero and nobody writes it: steps store states 1 .. S -- zeroed once with the buffer, not per pass)
        cw = B.get(tag + '/tf/cw', (S + 1, N, C), zero_on_alloc=True)
        hw = B.get(tag + '/tf/hw', (S + 1, N, C), zero_on_alloc=True)
        acts_w = B.get(tag + '/tf/acts_w', (S, N, G4))
        tmp_w = B.get(tag + '/tf/tmp_w', (N, G4))
        # bf16 arithmetic: every step also leaves its h as bf16 planes for the next one (two buffers in turn)
        fused = FUSED_STEP and N < UNFUSED_ROWS
        hpw = B.get(tag + '/tf/hpw', (2, hip.lstm_hplanes_floats(N, C)), zero_on_alloc=True) if fused and hip.lstm_bf(C, G4) else None
        for i in range(S):
            if fused:           # the step's GEMM and its gate math in one launch
                hip.lstm_step_fwd(hw[i], Kw[C:2 * C], G4, EW[i * N:(i + 1) * N], None, 1, mask[i], 1, cw[i], N, C, i > 0,
                                  cw[i + 1], hw[i + 1], acts_w[i], hp_in=None if hpw is None or i == 0 else hpw[(i - 1) & 1],
                                  hp_out=None if hpw is None else hpw[i & 1])
                continue
            hip.matmul(hw[i], Kw[C:2 * C], tmp_w)
            hip.call('ssc_lstm_pointwise_fwd', tmp_w, EW[i * N:(i + 1) * N], None, 1, mask[i], 1, cw[i], hw[i], N, C,
                     cw[i + 1], hw[i + 1], acts_w[i])
        lang = B.get(tag + '/tf/lang', (S * N, C))
        lang_ss = B.get(tag + '/tf/lang_ss', (S * N,))
        hip.call('ssc_row_l2norm_fwd', hw[1:], C, None, S * N, C, lang, lang_ss)
        Rall = B.get(tag + '/tf/Rall', (S * N, G4))
        hip.matmul(emb, Ka[C:2 * C], Rall)
        hip.matmul(lang, Ka[2 * C:3 * C], Rall, accumulate=True)
        return {'emb': emb, 'cw': cw, 'hw': hw, 'acts_w': acts_w, 'lang': lang, 'lang_ss': lang_ss, 'Rall': Rall}

    def start_words(self, text, C, tag, stream):
        """Run the image-independent part on ``stream`` now, so that its chain of small recurrent GEMMs overlaps the
        encoder convolutions; ``forward`` joins it.  Returns the ``prepare`` result to pass as ``text``.
        C: channels of the bottleneck feature (None: read off the cell's kernel, [2C, 4C])."""
        if C is None:
            C = self.s[self.pfx_w + 'kernel'].shape[1] // 4
        prep = text if isinstance(text, dict) else self.prepare(text, tag)
        if prep['S'] > 0 and stream is not None and 'words' not in prep:
            stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(stream):
                words = self._words_forward(prep, C, tag)
            prep = dict(prep, words=words, words_stream=stream)
        return prep

    def forward(self, e5, ab5, text, tag='g'):
        """e5 raw [N,h,w,C] (+ folded norm ab5), text int [N,T] on the HOST (or a ``prepare`` result)
        -> feat [N,h,w,C]."""
        s, B = self.s, self.b
        N, hh, ww, C = e5.shape
        P = hh * ww
        R = N * P
        prep = text if isinstance(text, dict) else self.prepare(text, tag)
        S = prep['S']
        assert prep['N'] == N
        ctx = {'N': N, 'P': P, 'C': C, 'S': S, 'tag': tag, 'shape': (N, hh, ww, C)}
        feat = B.get(tag + '/tf/feat', (N, hh, ww, C))
        if S == 0:      # every caption is all padding: relu(atanh(0)) = 0
            hip.fill(feat, 0.0)
            return feat, ctx
        tok, mask = prep['tok'], prep['mask']
        Ka, ba = s[self.pfx_a + 'kernel'], s[self.pfx_a + 'bias']
        G4 = 4 * C

        vis = B.get(tag + '/tf/vis', (R, C))
        vis_ss = B.get(tag + '/tf/vis_ss', (R,))
        hip.call('ssc_row_l2norm_fwd', e5, C, ab5, R, C, vis, vis_ss)
        Gv = B.get(tag + '/tf/Gv', (R, G4))
        hip.matmul(vis, Ka[0:C], Gv, bias=ba)
        words = prep.get('words')
        if words is None:
            words = self._words_forward(prep, C, tag)
        elif prep.get('words_stream') is not None:      # started earlier on a side stream (start_words): join it
            torch.cuda.current_stream().wait_stream(prep['words_stream'])
        emb, cw, hw, acts_w, lang, lang_ss, Rall = (words[k] for k in ('emb', 'cw', 'hw', 'acts_w', 'lang', 'lang_ss',
                                                                        'Rall'))

        # ---- multimodal LSTM (state per spatial position) ----
        ca = B.get(tag + '/tf/ca', (S + 1, R, C), zero_on_alloc=True)
        ha = B.get(tag + '/tf/ha', (S + 1, R, C), zero_on_alloc=True)
        acts_a = B.get(tag + '/tf/acts_a', (S, R, G4))
        tmp_a = B.get(tag + '/tf/tmp_a', (R, G4))
        fused = FUSED_STEP and R < UNFUSED_ROWS
        hpa = B.get(tag + '/tf/hpa', (2, hip.lstm_hplanes_floats(R, C)), zero_on_alloc=True) if fused and hip.lstm_bf(C, G4) else None
        for i in range(S):
            if fused:
                hip.lstm_step_fwd(ha[i], Ka[3 * C:4 * C], G4, Gv, Rall[i * N:(i + 1) * N], P, mask[i], P, ca[i], R, C, i > 0,
                                  ca[i + 1], ha[i + 1], acts_a[i], hp_in=None if hpa is None or i == 0 else hpa[(i - 1) & 1],
                                  hp_out=None if hpa is None else hpa[i & 1])
                continue
            if i == 0:
                hip.fill(tmp_a, 0.0)        # h_a = 0: skip the GEMM
            else:
                hip.matmul(ha[i], Ka[3 * C:4 * C], tmp_a)
            hip.call('ssc_lstm_pointwise_fwd', tmp_a, Gv, Rall[i * N:(i + 1) * N], P, mask[i], P, ca[i], ha[i], R, C,
                     ca[i + 1], ha[i + 1], acts_a[i])
        hip.call('ssc_squash_fwd', ha[S], R * C, feat)
        ctx.update(tok=tok, mask=mask, vis=vis, vis_ss=vis_ss, emb=emb, lang=lang, lang_ss=lang_ss, cw=cw, hw=hw,
                   acts_w=acts_w, ca=ca, ha=ha, acts_a=acts_a, feat=feat, Gv=Gv)
        return feat, ctx

    def join_backward(self):
        """Make the current stream wait for the word-branch gradients that ``backward`` left on a side stream."""
        if self._bwd_stream is not None:
            torch.cuda.current_stream().wait_stream(self._bwd_stream)
            self._bwd_stream = None

    _bwd_stream = None

    def backward(self, ctx, g_feat, side_stream=None):
        """g_feat [N,h,w,C]: gradient w.r.t. the fused feature.  Fills the TextLSTM parameter
        gradients and returns the gradient w.r.t. the *normalised* encoder_5 output [R,C]
        (None when no step ran).  With ``side_stream`` everything downstream of the multimodal BPTT that the image
        path does not need (word-term filter gradients, word LSTM BPTT, embedding gradient: a chain of small GEMMs)
        runs there; the caller must ``join_backward()`` before it reads the TextLSTM gradients."""
        s, B = self.s, self.b
        N, P, C, S, tag = ctx['N'], ctx['P'], ctx['C'], ctx['S'], ctx['tag']
        R, G4 = N * P, 4 * C
        gE, gKw, gbw = s.grad(self.emb_name), s.grad(self.pfx_w + 'kernel'), s.grad(self.pfx_w + 'bias')
        gKa, gba = s.grad(self.pfx_a + 'kernel'), s.grad(self.pfx_a + 'bias')
        if S == 0:
            for g in (gE, gKw, gbw, gKa, gba):
                hip.fill(g, 0.0)
            return None
        Kw, Ka = s[self.pfx_w + 'kernel'], s[self.pfx_a + 'kernel']
        ca, ha, acts_a, mask = ctx['ca'], ctx['ha'], ctx['acts_a'], ctx['mask']
        dh = B.get(tag + '/tfb/dh0', (R, C))
        dh2 = B.get(tag + '/tfb/dh1', (R, C))
        dc = B.get(tag + '/tfb/dc0', (R, C))
        dc2 = B.get(tag + '/tfb/dc1', (R, C))
        dg_all = B.get(tag + '/tfb/dg_all', (S, R, G4))      # gate gradients of every step: the filter gradient and the
        dGv = B.get(tag + '/tfb/dGv', (R, G4))              # per-sample row sums are batched over the steps afterwards
        dR = B.get(tag + '/tfb/dR', (S * N, G4))
        hip.call('ssc_squash_bwd', ha[S], ctx['feat'], g_feat, R * C, dh)
        hip.fill(dc, 0.0)
        hip.fill(dGv, 0.0)
        for i in range(S - 1, -1, -1):
            # the sequential chain: pointwise backward, then dh_{i-1} += dg_i Kh^T
            hip.call('ssc_lstm_pointwise_bwd', dh, dc, acts_a[i], ca[i], ca[i + 1], mask[i], P, R, C, dg_all[i], dc2, dh2,
                     dGv)
            if i > 0:       # h_a[0] = 0: nothing upstream of it
                hip.matmul_nt(dg_all[i], Ka[3 * C:4 * C], dh2, accumulate=True)
            dh, dh2 = dh2, dh
            dc, dc2 = dc2, dc
        dvis = B.get(tag + '/tfb/dvis', (R, C))
        hip.matmul_nt(dGv, Ka[0:C], dvis)
        dy5 = B.get(tag + '/tfb/dy5', (R, C))
        hip.call('ssc_row_l2norm_bwd', ctx['vis'], ctx['vis_ss'], dvis, R, C, dy5, 0)
        if side_stream is not None:
            side_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side_stream):
                self._backward_words(ctx, dR, dGv, gE, gKw, gbw, gKa, gba)
            self._bwd_stream = side_stream
        else:
            self._backward_words(ctx, dR, dGv, gE, gKw, gbw, gKa, gba)
        return dy5

    def _backward_words(self, ctx, dR, dGv, gE, gKw, gbw, gKa, gba):
        """Everything the image path does not wait for: the filter gradients of the multimodal cell (batched over the
        steps), the word-term gradients, the word LSTM BPTT and the embedding gradient."""
        s, B = self.s, self.b
        N, C, S, tag = ctx['N'], ctx['C'], ctx['S'], ctx['tag']
        G4 = 4 * C
        Kw, Ka = s[self.pfx_w + 'kernel'], s[self.pfx_a + 'kernel']
        mask = ctx['mask']
        P, R = ctx['P'], ctx['N'] * ctx['P']
        ha, dg_all = ctx['ha'], B.get(tag + '/tfb/dg_all', (S, R, G4))
        hip.call('ssc_group_rowsum', dGv, G4, 1, R, G4, gba, 0)
        hip.matmul_tn(ctx['vis'], dGv, gKa[0:C])
        # batched over the steps: dR_i = per-sample sums of dg_i over the positions; dKh = sum_i h_{i-1}^T dg_i
        hip.call('ssc_group_rowsum', dg_all, G4, S * N, P, G4, dR, 0)
        if S > 1:
            hip.matmul_tn(ha[1:S].reshape((S - 1) * R, C), dg_all[1:S].reshape((S - 1) * R, G4), gKa[3 * C:4 * C])
        else:
            hip.fill(gKa[3 * C:4 * C], 0.0)
        hip.matmul_tn(ctx['emb'], dR, gKa[C:2 * C])
        hip.matmul_tn(ctx['lang'], dR, gKa[2 * C:3 * C])
        demb = B.get(tag + '/tfb/demb', (S * N, C))
        hip.matmul_nt(dR, Ka[C:2 * C], demb)
        dlang = B.get(tag + '/tfb/dlang', (S * N, C))
        hip.matmul_nt(dR, Ka[2 * C:3 * C], dlang)
        dhw_ext = B.get(tag + '/tfb/dhw_ext', (S * N, C))
        hip.call('ssc_row_l2norm_bwd', ctx['lang'], ctx['lang_ss'], dlang, S * N, C, dhw_ext, 0)

        # ---- word LSTM BPTT ----
        cw, hw, acts_w = ctx['cw'], ctx['hw'], ctx['acts_w']
        dEW = B.get(tag + '/tfb/dEW', (S * N, G4))
        wh = B.get(tag + '/tfb/wdh0', (N, C))
        wh2 = B.get(tag + '/tfb/wdh1', (N, C))
        wc = B.get(tag + '/tfb/wdc0', (N, C))
        wc2 = B.get(tag + '/tfb/wdc1', (N, C))
        hip.fill(wh, 0.0)
        hip.fill(wc, 0.0)
        first = True
        for i in range(S - 1, -1, -1):
            hip.call('ssc_axpy', wh, dhw_ext[i * N:(i + 1) * N], 1.0, N * C)
            dgw = dEW[i * N:(i + 1) * N]
            hip.call('ssc_lstm_pointwise_bwd', wh, wc, acts_w[i], cw[i], cw[i + 1], mask[i], 1, N, C, dgw, wc2, wh2,
                     None)
            if i > 0:
                hip.matmul_tn(hw[i], dgw, gKw[C:2 * C], accumulate=not first)
                first = False
                hip.matmul_nt(dgw, Kw[C:2 * C], wh2, accumulate=True)
            wh, wh2 = wh2, wh
            wc, wc2 = wc2, wc
        if first:
            hip.fill(gKw[C:2 * C], 0.0)
        hip.matmul_tn(ctx['emb'], dEW, gKw[0:C])
        hip.matmul_nt(dEW, Kw[0:C], demb, accumulate=True)
        hip.call('ssc_group_rowsum', dEW, G4, 1, S * N, G4, gbw, 0)
        hip.fill(gE, 0.0)
        hip.call('ssc_embedding_scatter_add', gE, gE.shape[0], ctx['tok'], S * N, C, demb)
