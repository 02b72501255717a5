"""One tower of the reference training graph as eager HIP work:
build_single_graph + get_losses + optimize (graph_single.py:221-629).

D-step  = sess.run([opt_d, loss_d])  (main_procedure.py:202-216):
          G forward, D(real) + D(fake) forward, D backward (filter/norm grads), TF-Adam on D.
G-step  = sess.run([opt_g, loss_g, ...]) (:219-232):
          G forward, D(fake) forward, D data-gradient only, G backward, TF-Adam on G,
          spectral-norm ``u`` assignment (graph_single.py:178-210).

Data parallelism (graph_single.py:33-68, average_gradients): one process per GPU,
each rank = one tower with its own batch statistics; flat gradient buffers are
summed with RCCL all-reduce on a side HIP stream as soon as a contiguous section
of the backward pass has produced them, and the 1/world factor is folded into the
Adam kernel.
"""
import math
import os
import warnings

import numpy as np
import torch

from . import hip
from .hip import _dev_env
from .dist_utils import GradReducer
from .params import Buffers, ParamStore
from .pix2pix import Pix2PixDiscriminator, Pix2PixGenerator


class GanTrainer(object):
    """One tower for ``block_type`` 'Pix2Pix' or 'Residual' (graph_single.py:244-262 picks the generator /
    discriminator pair by block type; losses, optimizer and the step protocol are shared)."""

    def __init__(self, img=192, vocab_size=58, lstm_hybrid=True, lr_g=2e-4, lr_d=1e-4, max_iter_step=100000,
                 seed=0, sn=True, process_group=None, device='cuda', use_graphs=False, block_type='Pix2Pix',
                 segment_graphs=None, overlap_wgrad=None, optimizer='Adam', overlap_real=True, real_ahead=None):
        if not torch.cuda.is_available():
            raise RuntimeError('GanTrainer needs an MI355X (HIP) device: there is no CPU fallback')
        hip.lib()
        if not sn:
            # Config.sn = False switches the reference to another loss family (WGAN-GP / DRAGAN gradient penalty) and to
            # clip_by_global_norm + clip_by_norm before apply_gradients (graph_single.py:185-205, 355-386, 476-482).
            # Only the live branch (sn = True: softplus loss, spectral norm, no clipping) is built; running it with
            # spectral norm merely switched off would be a silently different algorithm.  (The reference cannot build that
            # branch either: both gradient-penalty losses call discriminator(interp, num_classes=...) without the
            # discrim_targets argument every discriminator requires -- graph_single.py:377-379, 457-459: a TypeError -- and
            # no CLI flag sets Config.sn.)
            raise NotImplementedError('Config.sn = False (gradient-penalty losses + gradient clipping) is not built')
        self.block_type = block_type
        self.store = ParamStore(block_type, vocab_size, img, device, seed)
        self.bufs = Buffers(device)
        if block_type == 'Pix2Pix':
            self.G = Pix2PixGenerator(self.store, self.bufs, lstm_hybrid)
            self.D = Pix2PixDiscriminator(self.store, self.bufs, sn)
        elif block_type == 'Residual':
            from .residual import ResidualDiscriminator, ResidualGenerator
            self.G = ResidualGenerator(self.store, self.bufs, 'fg', lstm_hybrid)
            self.D = ResidualDiscriminator(self.store, self.bufs, sn)
        elif block_type == 'MRU':
            from .mru import MRUDiscriminator, MRUGenerator
            self.G = MRUGenerator(self.store, self.bufs, lstm_hybrid)
            self.D = MRUDiscriminator(self.store, self.bufs, sn)
        else:
            raise NotImplementedError('training for block_type %r is not built' % block_type)
        self.lr_g, self.lr_d, self.max_iter_step = lr_g, lr_d, max_iter_step
        self.beta1, self.beta2, self.eps = 0.0, 0.9, 1e-8     # graph_single.py:588
        # get_optimizer (graph_single.py:584-593): Adam(beta1=0, beta2=0.9) is the CLI default; RMSProp(decay 0.9,
        # momentum 0, eps 1e-10), AdaGrad and AdaDelta (TF defaults) are selectable.  Slot 1 lives in scope.adam_v.
        self.optimizer = {'adam': 'adam', 'rmsprop': 'rmsprop', 'adagrad': 'adagrad', 'adadelta': 'adadelta'}.get(
            optimizer.lower())
        if self.optimizer is None:
            raise ValueError('unknown optimizer %r' % optimizer)
        for sc in (self.store.generator, self.store.discriminator):
            if self.optimizer == 'rmsprop':
                sc.adam_v.fill_(1.0)            # the rms slot starts at ones (tf.train.RMSPropOptimizer)
            elif self.optimizer == 'adagrad':
                sc.adam_v.fill_(0.1)            # initial_accumulator_value
            if self.optimizer in ('rmsprop', 'adadelta'):
                sc.adam_m = torch.zeros_like(sc.adam_v)
        self.loss = torch.zeros(2, dtype=torch.float64, device=device)   # [loss_g, loss_d], summed in double
        self.G.loss_acc, self.D.loss_acc = self.loss[0:1], self.loss[1:2]     # regularisers added inside backward
        self.reducer = GradReducer(process_group)
        self.world = self.reducer.world
        g = self.store.generator.offsets
        self._g_sections = self._sections(g, split_encoder_5=(block_type == 'Pix2Pix'))
        # discriminator (Pix2Pix pair): layer_4's filter is 8.4 of its 11.1 MB and sits, with layer_5 and the class head, at the
        # end of the flat buffer: final three layers before the backward pass ends, so that exchange runs beside them
        d = self.store.discriminator.offsets
        self._d_late = d['discriminator/layer_4/conv/filter'][0] if 'discriminator/layer_4/conv/filter' in d else None
        self._d_late_sent = False
        self._g_merge = int(os.environ.get('SSC_G_SECTIONS', '0')) if os.environ.get('SSC_G_SECTIONS', '0') in ('1', '2') else 0
        self._g_done = set()
        self._seg_eager_adam = os.environ.get('SSC_SEG_EAGER_ADAM', '1') == '1'
        self._sn_pending = None
        # hipGraph replay of whole D-/G-steps (the ~550 launches of a step are host-bound otherwise):
        # a step shape is run eagerly the first time, captured the second time, replayed afterwards
        self.use_graphs = bool(use_graphs)
        self._capturing = False
        self._graphs, self._seen, self._static = {}, set(), {}
        self._graph_gen = {}        # graph key -> hip.split_generation() at its capture
        # world > 1: collectives are NOT captured.  A step is captured as a chain of graph segments that end where
        # the backward pass hands a gradient section to the reducer; replay = segment, eager RCCL all-reduce on the
        # side stream, next segment, ...  (same overlap as eager mode, no dependence on graph-capturable RCCL).
        self.segment_graphs = (self.world > 1) if segment_graphs is None else bool(segment_graphs)
        # optional: filter gradients on a side stream, concurrent with the data-gradient chain (hip.WGRAD_STREAM;
        # Pix2Pix pair only -- its backward touches a filter gradient again only at the section joins below).
        # OFF by default: measured 25.39 vs 24.77 ms/step (batch 32) -- both kernel families already fill the CUs
        # (66 KB LDS per workgroup), so co-scheduling them only adds contention.
        if overlap_wgrad is None:
            overlap_wgrad = os.environ.get('SSC_OVERLAP_WGRAD', '0') == '1'
        if overlap_wgrad and self.use_graphs and overlap_wgrad != 'force':      # 'force': tests/test_gpu_stream_hazards.py
            # round 4: with several filter gradients queued on the side stream the CAPTURED step is not bit-identical to the eager
            # one (a hazard that stream order hides in eager mode; profiles/NOTEBOOK_r04.md section 8): eager steps only
            import warnings as _w
            _w.warn('SSC_OVERLAP_WGRAD=1 is ignored when steps are captured into hipGraphs')
            overlap_wgrad = False
        self._wgrad_stream = torch.cuda.Stream() if (overlap_wgrad and block_type == 'Pix2Pix') else None
        self._aux_stream = torch.cuda.Stream() if overlap_real else None
        # discriminator backward of the real and of the fake pair side by side (Pix2Pix / Residual discriminators: their
        # backward touches a gradient only through store.grad(); the MRU one keeps per-call spectral-norm state)
        # Not together with the filter-gradient side stream: both passes' filter gradients would then land on
        # hip.WGRAD_STREAM, which neither pass's stream waits for before the two buffers are added.
        self._dbwd_concurrent = (overlap_real and block_type in ('Pix2Pix', 'Residual') and self._wgrad_stream is None and
                                 os.environ.get('SSC_DBWD_CONCURRENT', '1') == '1')
        self._text_stream = torch.cuda.Stream() if (overlap_real and os.environ.get('SSC_TEXT_STREAM', '1') == '1') else None
        # Pix2Pix pair with its chains side by side (two discriminator passes, run-ahead generator forward, real pass ahead, held
        # filter gradients): the conv launches take the small-LDS form (ssc_conv_desc.lds_hint) -- 13.0 -> 12.67 ms per iteration
        # on one box although the kernel alone is 3-5 % slower.  MRU (one chain most of the time) loses 2 % with it: off there.
        _cr = os.environ.get('SSC_CO_RUN', '1')        # 0: never, 1: the Pix2Pix pair, all: every block type (A/B)
        self._co_run = bool(overlap_real and ((block_type == 'Pix2Pix' and _cr == '1') or _cr == 'all'))
        # generator forward of the next G-step inside the D-step (train_iteration)
        self.run_ahead = (overlap_real and os.environ.get('SSC_RUN_AHEAD', '1') == '1')
        self._ahead_stream = torch.cuda.Stream() if self.run_ahead else None
        self._ahead = None
        self._ahead_pending = False
        # ... and, the other way round, the REAL half of the next D-step inside the G-step: D(real) forward + backward depend on
        # the discriminator's variables (untouched by a G-step), the next real batch and the spectral-norm u the G-step is
        # about to assign -- not on the generator.  The G-step is one dependent chain (D forward -> losses -> D data gradient
        # -> G backward, with ~12 tiny launches at each transition and the caption branch's BPTT chain in the middle:
        # scripts/timeline_dump.py shows ~1.5 ms of its 7 ms with no full-size launch in flight); the D-step already runs
        # three chains.  Moving 3.3 ms of independent full-size work under the G-step does fill those holes -- and empties the
        # D-step of the co-running chain that filled ITS launch tails: measured (scripts/branch_marks.py) D-step 10.45 -> 7.53
        # ms, G-step 6.97 -> 9.90 ms, the iteration 17.42 -> 17.43 ms; 1798 vs 1813 images/s over three interleaved runs.  Two
        # full-size chains side by side run at the speed of one after the other wherever they meet (the launches fill the LDS
        # of every CU on their own), so work moved between the steps is zero-sum.  That was the exact-fp32 era (66-101 KB of LDS
        # per workgroup).  Round 5: with the bf16-split kernels and the filter gradient on ONE 51 KB stage a filter-gradient
        # workgroup fits beside a conv workgroup of another chain, and the real pass inside the G-step no longer loses: 13.14 -> 12.93 ms
        # per iteration on one box (2436 -> 2475 images/s, three interleaved runs each, every pair faster), 13.29 vs 13.29 on a
        # second one (profiles/NOTEBOOK_r05.md section 11).  ON by default (SSC_REAL_AHEAD=0 /
        # real_ahead=False: off); tested bit for bit against the in-line trainer at full size.  Pix2Pix pair, one GPU (a fork
        # may not cross the end of a graph segment, and with world > 1 the G-step is cut at every gradient section).
        if real_ahead is None:
            real_ahead = os.environ.get('SSC_REAL_AHEAD', '1') == '1'
        self.real_ahead = bool(real_ahead and overlap_real and block_type == 'Pix2Pix' and self._dbwd_concurrent and
                               not self.segment_graphs)
        self._real_stream = torch.cuda.Stream() if self.real_ahead else None
        self._real = None               # what the run-ahead real pass left for the D-step: {'sn', 'cr'}
        self._real_pending, self._real_key = False, None
        self.loss_real = torch.zeros(1, dtype=torch.float64, device=device)      # its loss terms
        self.use_graphs_infer = os.environ.get('SSC_INFER_GRAPHS', '1') == '1'   # hipGraph replay of generate / generate_u8
        self.G.text_stream = self._text_stream          # forward half: every generator
        if block_type == 'Pix2Pix':
            self.G.text_stream_bwd = None if self.segment_graphs else self._text_stream
            # (graph segments, many towers: the word half runs in line -- a fork may not cross a segment end; starting it after the
            # decoders' hand-over and joining it after encoder_4's backward measured slower, profiles/NOTEBOOK_r04.md)
            # norm backward of a layer next to the filter gradient of the layer above (Pix2PixGenerator._fork).  OFF: measured
            # 18.28 vs 18.07 ms/step -- a fork + join per layer costs more in cross-queue dependencies of the replayed
            # graph than the three small launches it hides
            if overlap_real and _dev_env('SSC_BN_OVERLAP', '0') == '1':
                self.G.bn_stream = torch.cuda.Stream()
        self._seg = None
        self.lr_dev = torch.zeros(2, dtype=torch.float32, device=device)     # Adam step sizes [G, D]

    # ------------------------------------------------------------------ helpers
    @staticmethod
    def _sections(offsets, split_encoder_5=False):
        """Contiguous flat ranges in the order the generator backward finishes them."""
        names = list(offsets.keys())
        i_emb = names.index('generator/TextLSTM/embedding')
        i_fc = names.index('generator/fully_connected/weights')
        end = offsets[names[-1]][0] + (offsets[names[-1]][1] + 63) // 64 * 64
        o_emb, o_fc = offsets[names[i_emb]][0], offsets[names[i_fc]][0]
        sec = {'decoders': (o_fc, end), 'text': (o_emb, o_fc), 'encoders': (0, o_emb)}
        # Pix2Pix: encoder_5 (17 of the encoders' 28 MB) is final one layer into the encoder backward, and it sits last among
        # the encoders in the flat buffer: its all-reduce starts then, beside the backward of encoder_4..1, and only the
        # remaining 11 MB are exchanged after the backward pass has ended
        e5 = 'generator/encoder_5/conv/filter'
        if split_encoder_5 and e5 in offsets and names.index(e5) < i_emb and all(n.startswith('generator/encoder_5/') for n in names[names.index(e5):i_emb]):
            sec['encoder_5'] = (offsets[e5][0], o_emb)
            sec['encoders'] = (0, offsets[e5][0])
        return sec

    def decay(self, counter):
        """graph_single.py:139 in fp32: max(0.2, 1 - counter/max_iter*0.9)."""
        c = np.float32(counter) / np.float32(self.max_iter_step) * np.float32(0.9)
        return float(max(np.float32(0.2), np.float32(1.0) - c))

    def _allreduce_async(self, flat, lo, hi):
        if self._seg is not None:
            self._seg_break(('reduce', flat, lo, hi))
        else:
            self.reducer.reduce_async(flat, lo, hi)

    def _allreduce_wait(self):
        if self._seg is not None:
            self._seg_break(('wait',))
        else:
            self.reducer.wait()

    # ------------------------------------------------------------------ segmented capture
    def _seg_begin_graph(self):
        g = hip.new_graph()
        # thread_local: the RCCL watchdog thread polls events while we capture; under the default 'global' mode that
        # call invalidates the capture (hipErrorStreamCaptureInvalidated, seen intermittently)
        g.capture_begin(pool=self._seg['pool'], capture_error_mode='thread_local')
        self._seg['cur'], self._seg['mark'] = g, hip.LAUNCHES

    def _seg_end_graph(self):
        sg = self._seg
        with warnings.catch_warnings():     # "The CUDA Graph is empty" for a segment between two back-to-back collectives
            warnings.simplefilter('ignore')
            sg['cur'].capture_end()
        if hip.LAUNCHES != sg['mark']:          # segments without a single launch are dropped
            sg['ops'].append(('graph', sg['cur']))
        sg['cur'] = None

    def _seg_break(self, op):
        self._seg_end_graph()
        self._seg['ops'].append(op)
        self._seg_begin_graph()

    def _capture_segments(self, impl, sbatch):
        stream = torch.cuda.Stream()
        stream.wait_stream(torch.cuda.current_stream())
        self._seg = {'ops': [], 'pool': torch.cuda.graph_pool_handle(), 'cur': None, 'mark': 0}
        self._capturing = True
        try:
            with torch.cuda.stream(stream):
                self._seg_begin_graph()
                impl(sbatch)
                self._seg_end_graph()
        except Exception:
            cur = self._seg.get('cur') if self._seg else None
            if cur is not None:
                try:
                    cur.capture_end()
                except Exception:
                    pass
            raise
        finally:
            self._capturing = False
            ops, self._seg = (self._seg['ops'] if self._seg else []), None
        torch.cuda.current_stream().wait_stream(stream)
        return ops

    def _replay_segments(self, ops):
        for op in ops:
            if op[0] == 'graph':
                op[1].replay()
            elif op[0] == 'reduce':
                self.reducer.reduce_async(op[1], op[2], op[3])
            elif op[0] == 'adam':
                self._adam_launch_now(op[1], op[2])
            else:
                self.reducer.wait()

    def _adam_prepare(self, scope, idx, lr):
        """Host part of the optimizer step: advance t, put lr_t = lr*sqrt(1-b2^t)/(1-b1^t) in device memory."""
        scope.adam_t += 1
        t = scope.adam_t
        lr_t = lr * math.sqrt(1.0 - self.beta2 ** t) / (1.0 - self.beta1 ** t) if self.optimizer == 'adam' else lr
        self.lr_dev[idx:idx + 1].fill_(float(lr_t))

    def _adam_launch(self, scope, idx):
        if self._seg is not None and self._seg_eager_adam:
            # segmented capture: the optimizer launch that follows the last exchange is issued eagerly at replay instead of
            # becoming a one-kernel graph segment of its own (a graph launch costs far more than the launch it would hold)
            self._seg_end_graph()
            self._seg['ops'].append(('adam', scope, idx))
            self._seg_begin_graph()
            return
        self._adam_launch_now(scope, idx)

    def _adam_launch_now(self, scope, idx):
        lr_dev, gs = self.lr_dev[idx:idx + 1], 1.0 / self.world
        if self.optimizer == 'adam':
            hip.call('ssc_adam_tf', scope.flat, scope.grad, None, scope.adam_v, scope.numel, 0.0, lr_dev, self.beta1,
                     self.beta2, self.eps, gs)
        elif self.optimizer == 'rmsprop':
            hip.call('ssc_optimizer_step', 1, scope.flat, scope.grad, scope.adam_v, scope.adam_m, scope.numel, lr_dev,
                     0.9, 0.0, 1e-10, gs)
        elif self.optimizer == 'adagrad':
            hip.call('ssc_optimizer_step', 2, scope.flat, scope.grad, scope.adam_v, None, scope.numel, lr_dev, 0.0, 0.0,
                     0.0, gs)
        else:
            hip.call('ssc_optimizer_step', 3, scope.flat, scope.grad, scope.adam_v, scope.adam_m, scope.numel, lr_dev,
                     0.95, 0.0, 1e-8, gs)
        hip.refresh_splits(scope.flat)      # the bf16 planes of this scope's filters follow the weights

    def _static_inputs(self, kind, batch):
        N, _, H, W = batch['sketches'].shape
        skey = (kind, N, H, W)
        st = self._static.get(skey)
        if st is None:
            st = {k: torch.empty_like(v) for k, v in batch.items() if isinstance(v, torch.Tensor)}
            self._static[skey] = st
        return st

    def input_buffers(self, kind, like):
        """The device tensors a replayed 'd' / 'g' step reads its inputs from (graphs replay fixed addresses), allocated
        after ``like`` and filled with it.  An input pipeline that writes the next batch straight into them -- and passes them as the step's
        batch, with the host-side ``text`` -- saves the per-step device copies; any other batch is copied in."""
        st = self._static_inputs(kind, like)
        for k, v in st.items():
            if like[k].data_ptr() != v.data_ptr():
                v.copy_(like[k])
        out = dict(like)
        out.update(st)
        if self.G.lstm_hybrid and not isinstance(like['text'], dict):
            # the caption tokens too: time-major ids and skip mask on the device (own buffers per step kind)
            out['text'] = self.G.text.prepare(like['text'], 'g' + kind)
        return out

    def _static_batch(self, kind, batch):
        """The step's inputs in the tensors a replayed graph reads (copied in unless they already are those tensors)."""
        st = self._static_inputs(kind, batch)
        for k, v in st.items():
            if batch[k].data_ptr() != v.data_ptr():     # an input pipeline may fill the static buffers itself
                v.copy_(batch[k])
        sbatch = dict(st)
        if self.G.lstm_hybrid and not isinstance(batch['text'], dict):      # a dict: already prepared (input_buffers)
            sbatch['text'] = self.G.text.prepare(batch['text'], 'g' + kind)
        else:
            sbatch['text'] = batch['text']
        return sbatch

    def _run_step(self, kind, batch, counter, ahead=None, use_ahead=False, real=None, use_real=False):
        """Eager, capture or replay of one D-/G-step.  ``ahead``: the NEXT generator step's batch, whose generator
        forward this discriminator step also runs (on a side stream); ``use_ahead``: this generator step starts from it.
        ``real``: the NEXT discriminator step's batch, whose real pass this generator step also runs; ``use_real``: this
        discriminator step starts from it."""
        keep = hip.CO_RUN
        hip.CO_RUN = self._co_run       # (read when a launch descriptor is filled: eager steps and captures)
        try:
            return self._run_step_inner(kind, batch, counter, ahead, use_ahead, real, use_real)
        finally:
            hip.CO_RUN = keep

    def _run_step_inner(self, kind, batch, counter, ahead, use_ahead, real, use_real):
        scope, idx, lr = ((self.store.discriminator, 1, self.lr_d) if kind == 'd' else
                          (self.store.generator, 0, self.lr_g))
        if kind == 'd':
            impl = lambda b: self._d_impl(b, ahead, use_real)
        else:
            impl = lambda b: self._g_impl(b, use_ahead, real)
        if not self.use_graphs or hip.PROFILE is not None:
            self._adam_prepare(scope, idx, lr * self.decay(counter))
            return impl(batch)
        # static inputs: graphs replay fixed device addresses
        N, _, H, W = batch['sketches'].shape
        skey = (kind, N, H, W)
        sbatch = self._static_batch(kind, batch)
        S = sbatch['text']['S'] if isinstance(sbatch['text'], dict) else -1
        key = skey + (S,)
        if ahead is not None:           # the generator step's inputs must be in place before this graph reads them
            ahead = self._static_batch('g', ahead)
            key = key + ('ahead', ahead['text']['S'] if isinstance(ahead['text'], dict) else -1)
        elif use_ahead:
            key = key + ('use_ahead',)
        if real is not None:            # likewise the next discriminator step's inputs (its graph will find them in place)
            real = self._static_batch('d', real)
            key = key + ('real',)
        elif use_real:
            key = key + ('use_real',)
        if kind == 'd':
            impl = lambda b, a=ahead: self._d_impl(b, a, use_real)
        else:
            impl = lambda b, r=real: self._g_impl(b, use_ahead, r)
        self._adam_prepare(scope, idx, lr * self.decay(counter))
        g = self._graphs.get(key)
        if g is None:
            if key not in self._seen:       # first time: eager (allocates buffers, sets kernel attributes)
                self._seen.add(key)
                return impl(sbatch)
            try:
                if self.segment_graphs:
                    g = self._capture_segments(impl, sbatch)
                else:
                    g = hip.new_graph()
                    self._capturing = True
                    try:
                        with torch.cuda.graph(g, capture_error_mode='thread_local'):
                            impl(sbatch)
                    finally:
                        self._capturing = False
            except Exception as e:      # never lose a training run to graph capture: fall back to eager launches
                print('hipGraph capture failed (%r): continuing with eager launches' % (e,))
                self.use_graphs = False
                self._seg = None
                self._g_done, self._d_late_sent = set(), False      # nothing of the abandoned capture was exchanged
                torch.cuda.synchronize()
                scope.adam_t -= 1       # _adam_prepare ran once for this step already
                self._adam_prepare(scope, idx, lr * self.decay(counter))
                return impl(sbatch)
            self._graphs[key] = g
            self._graph_gen[key] = hip.split_generation()
        # bf16 planes and replayed graphs (hip.resplit_stale / refresh_new_splits): weights replaced through torch since the
        # last launch are split again in front of the replay; filters that met their first bf16 launch after this graph was
        # captured are not in its optimizer refresh and are split behind it
        hip.resplit_stale()
        if isinstance(g, list):
            self._replay_segments(g)
        else:
            g.replay()
        hip.refresh_new_splits(scope.flat, self._graph_gen[key])
        return self.loss[1:2] if kind == 'd' else self.loss[0:1]

    def _g_forward(self, batch, tag='g', **kw):
        if self.block_type == 'MRU':        # class-conditional norms (models_collection.py:80-82, 270-272)
            return self.G.forward(batch['sketches'], batch['text'], batch['class_id'], batch['noise_vec'], tag, **kw)
        return self.G.forward(batch['sketches'], batch['text'], batch['noise_vec'], tag, **kw)

    def _pack_fake(self, batch, tag='g'):
        B = self.bufs
        N, _, H, W = batch['sketches'].shape
        xd_f = B.get('xd_fake' if tag == 'g' else 'xd_fake_' + tag, (N, H, W, 8), zero_on_alloc=True)
        hip.nchw_to_nhwc(batch['sketches'], xd_f, 0)
        gctx = self._g_forward(batch, tag, out=xd_f, out_coff=3)
        return xd_f, gctx

    # ------------------------------------------------------------------ steps
    @staticmethod
    def _batch_key(batch):
        return (batch['images_d'].data_ptr(), batch['sketches'].data_ptr(), batch['class_id_d'].data_ptr())

    def d_step(self, batch, counter=0, ahead=None, use_real=False):
        """One discriminator update; returns the device scalar loss_d (a view of self.loss).
        ahead: the batch of the generator step that follows -- its generator forward then runs inside this step
        (``run_ahead``; call ``g_step(that batch, use_ahead=True)`` next).
        use_real: the preceding ``g_step(..., next_d=batch)`` ran this step's real pass (``real_ahead``); taken only if
        ``batch`` is the batch that call was given (same tensors)."""
        use_real = bool(use_real and self._real_pending and hip.PROFILE is None and self._real is not None and
                        self._real_key == self._batch_key(batch))
        self._real_pending = False
        if ahead is not None and self.run_ahead and hip.PROFILE is None:
            self._ahead_pending = True
            return self._run_step('d', batch, counter, ahead=ahead, use_real=use_real)
        self._ahead_pending = False
        return self._run_step('d', batch, counter, use_real=use_real)

    def _d_impl(self, batch, ahead=None, use_real=False):
        if ahead is not None:
            # The generator does not change during a discriminator step, so the generator forward of the generator step
            # that follows can run now, on its own stream, in whatever the discriminator step leaves idle (launch tails,
            # partly filled rounds).  No nested fork: the word half of its caption branch stays in line.
            main = torch.cuda.current_stream()
            hip.mark('d/start')
            self._ahead_stream.wait_stream(main)
            with torch.cuda.stream(self._ahead_stream):
                ts, self.G.text_stream = self.G.text_stream, None
                try:
                    hip.mark('d/ahead G forward: first')
                    self._ahead = self._pack_fake(ahead, 'ga')
                    hip.mark('d/ahead G forward: last')
                finally:
                    self.G.text_stream = ts
        self._ahead_forked = ahead is not None
        loss_d = self.d_gradients(batch, use_real)
        if ahead is not None:       # joined before the gradient all-reduce: a fork may not cross the end of a graph segment
            torch.cuda.current_stream().wait_stream(self._ahead_stream)
        self._apply_d_launch()
        hip.mark('d/end')
        return loss_d

    def apply_d(self, counter=0):
        """optim_d.apply_gradients on the (all-reduced) discriminator gradients."""
        self._adam_prepare(self.store.discriminator, 1, self.lr_d * self.decay(counter))
        self._apply_d_launch()

    def _apply_d_launch(self):
        sc = self.store.discriminator
        # (the late section -- layer_4, layer_5, class head -- may already be on its way: _d_gradients)
        self._allreduce_async(sc.grad, 0, self._d_late if self._d_late_sent else sc.numel)
        self._d_late_sent = False
        self._allreduce_wait()
        self._adam_launch(sc, 1)

    def allreduce_plan(self):
        """Bytes per iteration and per exchange: what a SCALE line of bench.py can be checked against."""
        gs = {k: 4 * (hi - lo) for k, (lo, hi) in self._g_sections.items()}
        dn = 4 * self.store.discriminator.numel
        ds = {'all': dn} if self._d_late is None or self.block_type != 'Pix2Pix' or self.world == 1 else \
            {'layer_4 + layer_5 + class head': dn - 4 * self._d_late, 'layer_1..3': 4 * self._d_late}
        return {'generator_sections_bytes': gs, 'discriminator_sections_bytes': ds,
                'bytes_per_iteration': sum(gs.values()) + sum(ds.values()), 'collective': 'sum all-reduce, fp32'}

    def d_gradients(self, batch, use_real=False):
        """loss_d and d loss_d / d discriminator variables (compute_gradients, graph_single.py:309-312).
        With more than one tower (Pix2Pix pair) the call also STARTS the all-reduce of the late section of the flat gradient
        buffer (layer_4, layer_5, class head) beside the rest of the backward pass: until ``apply_d`` has waited for it that
        section of ``store.discriminator.grad`` is in flight -- read gradients after ``apply_d``, or on one tower."""
        if self._d_late_sent:
            # a previous call already sent the late section on its way and nobody applied it (apply_d was never called):
            # let that exchange finish before the buffer is written again, and start over
            self._allreduce_wait()
            self._d_late_sent = False
        hip.WGRAD_STREAM = self._wgrad_stream
        try:
            return self._d_gradients_fake_only(batch) if use_real else self._d_gradients(batch)
        finally:
            hip.join_wgrad()
            hip.WGRAD_STREAM = None

    def _d_real_pass(self, batch, u=None):
        """The real half of a discriminator step: D(real) forward, its loss terms (into ``loss_real``) and its backward into the
        discriminator's gradient buffer.  Runs on the current stream; ``u``: spectral-norm vector to start from (inside a
        generator step: the u that step will assign at its end)."""
        B = self.bufs
        N, _, H, W = batch['sketches'].shape
        sn = self.D.prepare_sn(tag='d/sn_r', u=u)
        xd_r = B.get('xd_real', (N, H, W, 8), zero_on_alloc=True)
        hip.nchw_to_nhwc(batch['sketches'], xd_r, 0)
        hip.nchw_to_nhwc(batch['images_d'], xd_r, 3)
        cr = self.D.forward(xd_r, sn, 'dr')
        lr = self.loss_real
        lr.zero_()
        rows = cr['disc'].shape[0] * cr['disc'].shape[1] * cr['disc'].shape[2]
        dl5_r = B.get('dl5_r', cr['disc'].shape, zero_on_alloc=True)
        hip.call('ssc_softplus_loss', cr['disc'], 4, rows, -1.0, 1.0 / rows, lr, dl5_r, 1.0 / rows)
        K = cr['logits'].shape[1]
        dlog_r = B.get('dlog_r', (N, K))
        hip.call('ssc_acgan_loss', cr['logits'], batch['class_id_d'], N, K, 1, 1.0, lr, dlog_r)
        self.D.backward(cr, dl5_r, dlog_r, sn, True, False, accumulate=False)
        return {'sn': sn, 'cr': cr}

    def _d_gradients_fake_only(self, batch):
        """d_gradients when the real pass of this step ran ahead (inside the preceding generator step): generator forward,
        D(fake) forward + backward into the second gradient buffer, the two buffers added."""
        B, s = self.bufs, self.store
        sn, cr = self._real['sn'], self._real['cr']
        xd_f, gctx = self._pack_fake(batch)
        cf = self.D.forward(xd_f, sn, 'df')
        loss_d = self.loss[1:2]
        loss_d.zero_()
        rows = cf['disc'].shape[0] * cf['disc'].shape[1] * cf['disc'].shape[2]
        dl5_f = B.get('dl5_f', cf['disc'].shape, zero_on_alloc=True)
        hip.call('ssc_softplus_loss', cf['disc'], 4, rows, 1.0, 1.0 / rows, loss_d, dl5_f, 1.0 / rows)
        loss_d.add_(self.loss_real)
        sc = s.discriminator
        if getattr(sc, 'grad2', None) is None:
            sc.grad2 = torch.zeros_like(sc.grad)
            sc.g2 = type(sc.g)((n, sc.grad2[o:o + k].view(shp)) for n, (o, k, shp) in sc.offsets.items())
        sc.g, sc.g2 = sc.g2, sc.g
        try:
            self.D.backward(cf, dl5_f, None, sn, True, False, accumulate=False)
        finally:
            sc.g, sc.g2 = sc.g2, sc.g
        hip.call('ssc_axpy', sc.grad, sc.grad2, 1.0, sc.numel)
        self.D.finish_sn_backward(sn)
        hip.call('ssc_l2_reg', s['discriminator/fully_connected/weights'],
                 s['discriminator/fully_connected/weights'].numel(), 1e-6, loss_d,
                 s.grad('discriminator/fully_connected/weights'))
        self.last = {'gctx': gctx, 'cr': cr, 'cf': cf}
        return loss_d

    def _d_gradients(self, batch):
        B, s = self.bufs, self.store
        N, _, H, W = batch['sketches'].shape
        xd_r = B.get('xd_real', (N, H, W, 8), zero_on_alloc=True)

        def real_branch(sn):
            hip.nchw_to_nhwc(batch['sketches'], xd_r, 0)
            hip.nchw_to_nhwc(batch['images_d'], xd_r, 3)
            return self.D.forward(xd_r, sn, 'dr')

        if self._aux_stream is not None and hip.PROFILE is None:
            # D(real) does not depend on the generator: run it (and the spectral-norm power iterations in front of it) on a
            # second stream so that its full-size launches fill the CUs the generator's caption branch (a chain of small
            # GEMMs) leaves idle
            self._aux_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._aux_stream):
                hip.mark('d/D(real) forward: first')
                sn = self.D.prepare_sn()
                cr = real_branch(sn)
                hip.mark('d/D(real) forward: last')
            hip.mark('d/G forward: first')
            xd_f, gctx = self._pack_fake(batch)
            hip.mark('d/G forward: last')
            torch.cuda.current_stream().wait_stream(self._aux_stream)
        else:
            sn = self.D.prepare_sn()
            xd_f, gctx = self._pack_fake(batch)
            cr = real_branch(sn)
        cf = self.D.forward(xd_f, sn, 'df')
        hip.mark('d/D(fake) forward: last')
        loss_d = self.loss[1:2]
        loss_d.zero_()
        rows = cr['disc'].shape[0] * cr['disc'].shape[1] * cr['disc'].shape[2]
        dl5_r = B.get('dl5_r', cr['disc'].shape, zero_on_alloc=True)
        dl5_f = B.get('dl5_f', cf['disc'].shape, zero_on_alloc=True)
        hip.call('ssc_softplus_loss', cf['disc'], 4, rows, 1.0, 1.0 / rows, loss_d, dl5_f, 1.0 / rows)
        hip.call('ssc_softplus_loss', cr['disc'], 4, rows, -1.0, 1.0 / rows, loss_d, dl5_r, 1.0 / rows)
        K = cr['logits'].shape[1]
        dlog_r = B.get('dlog_r', (N, K))
        hip.call('ssc_acgan_loss', cr['logits'], batch['class_id_d'], N, K, 1, 1.0, loss_d, dlog_r)
        if self._dbwd_concurrent and self._aux_stream is not None and hip.PROFILE is None:
            # The two backward passes of the discriminator step (real pair, fake pair) share nothing but the filters they
            # read: run them side by side -- the fake pair on the second stream into a second gradient buffer, the real
            # pair in line -- so that each chain's launch tails and partly filled rounds are filled by the other, then add
            # the buffers.  The fake pair's pass goes to the second buffer: it touches a subset of the variables (no class
            # head), the rest of that buffer stays zero from its allocation.  (17.42 vs 17.70 ms per iteration in line.)
            sc = s.discriminator
            if getattr(sc, 'grad2', None) is None:
                sc.grad2 = torch.zeros_like(sc.grad)
                sc.g2 = type(sc.g)((n, sc.grad2[o:o + k].view(shp)) for n, (o, k, shp) in sc.offsets.items())
            main = torch.cuda.current_stream()

            def both(**kw):
                """One stretch of the two passes side by side; returns what each returned."""
                self._aux_stream.wait_stream(main)
                with torch.cuda.stream(self._aux_stream):
                    sc.g, sc.g2 = sc.g2, sc.g
                    try:
                        hip.mark('d/D backward (fake): first')
                        rf = self.D.backward(cf, dl5_f, None, sn, True, False, accumulate=False, **kw.get('fake', {}))
                        hip.mark('d/D backward (fake): last')
                    finally:
                        sc.g, sc.g2 = sc.g2, sc.g
                hip.mark('d/D backward (real): first')
                rr = self.D.backward(cr, dl5_r, dlog_r, sn, True, False, accumulate=False, **kw.get('real', {}))
                hip.mark('d/D backward (real): last')
                main.wait_stream(self._aux_stream)
                hip.join_wgrad()    # (no side-stream filter gradients in this mode; kept so that the add can never run early)
                return rr, rf

            late = self._d_late if (self.world > 1 and self.block_type == 'Pix2Pix') else None
            if late is None:
                both()
                hip.call('ssc_axpy', sc.grad, sc.grad2, 1.0, sc.numel)
            else:
                # more than one tower: the two passes side by side in two stretches.  After the first (layer_5, the class head,
                # layer_4 and the data gradient into layer_3) the late section of the flat buffer -- layer_4's filter (8.4 of
                # the 11.1 MB), layer_5, the class head -- is complete in both buffers: added, and on its way to the all-reduce
                # beside the second stretch (layers 3..1 of both passes).  A fork may not cross the end of a graph segment, so
                # the two chains meet at that point; they are the same work on two batches and arrive together.
                cont_r, cont_f = both(real={'stop_after': 4}, fake={'stop_after': 4})
                hip.call('ssc_axpy', sc.grad[late:], sc.grad2[late:], 1.0, sc.numel - late)
                if getattr(self, '_ahead_forked', False):
                    torch.cuda.current_stream().wait_stream(self._ahead_stream)
                self.D.finish_sn_backward(sn)
                hip.call('ssc_l2_reg', s['discriminator/fully_connected/weights'],
                         s['discriminator/fully_connected/weights'].numel(), 1e-6, loss_d,
                         s.grad('discriminator/fully_connected/weights'))
                self._allreduce_async(sc.grad, late, sc.numel)
                self._d_late_sent = True
                both(real={'resume': cont_r}, fake={'resume': cont_f})
                hip.call('ssc_axpy', sc.grad, sc.grad2, 1.0, late)
                self.last = {'gctx': gctx, 'cr': cr, 'cf': cf}
                return loss_d
        else:
            self.D.backward(cr, dl5_r, dlog_r, sn, True, False, accumulate=False)
            if self.world > 1 and self.block_type == 'Pix2Pix' and self._d_late is not None:
                # more than one tower: the two passes in line, and the late section of the flat buffer -- layer_4's filter
                # (8.4 of the 11.1 MB), layer_5, the class head -- goes to the all-reduce as soon as the second pass has
                # added its layer_4 gradient, beside the backward of layers 3..1 (the class head's gradient is complete after
                # the real pass: the fake pair has no class term)
                sc = s.discriminator

                def late(k):
                    if k == 4:
                        hip.join_wgrad()
                        if getattr(self, '_ahead_forked', False):       # a fork may not cross the end of a graph segment
                            torch.cuda.current_stream().wait_stream(self._ahead_stream)
                        self.D.finish_sn_backward(sn)
                        hip.call('ssc_l2_reg', s['discriminator/fully_connected/weights'],
                                 s['discriminator/fully_connected/weights'].numel(), 1e-6, loss_d,
                                 s.grad('discriminator/fully_connected/weights'))
                        self._allreduce_async(sc.grad, self._d_late, sc.numel)
                        self._d_late_sent = True

                self.D.backward(cf, dl5_f, None, sn, True, False, accumulate=True, after_layer=late)
                hip.join_wgrad()
                self.last = {'gctx': gctx, 'cr': cr, 'cf': cf}
                return loss_d
            self.D.backward(cf, dl5_f, None, sn, True, False, accumulate=True)
        hip.join_wgrad()
        self.D.finish_sn_backward(sn)
        hip.call('ssc_l2_reg', s['discriminator/fully_connected/weights'],
                 s['discriminator/fully_connected/weights'].numel(), 1e-6, loss_d,
                 s.grad('discriminator/fully_connected/weights'))
        self.last = {'gctx': gctx, 'cr': cr, 'cf': cf}
        return loss_d

    def g_step(self, batch, counter=0, use_ahead=False, next_d=None):
        """One generator update (+ spectral-norm u assignment); returns the device scalar loss_g.
        use_ahead: start from the forward pass the preceding ``d_step(..., ahead=batch)`` ran for this batch.
        next_d: the batch of the discriminator step that follows -- its real pass then runs inside this step
        (``real_ahead``; call ``d_step(that batch, use_real=True)`` next)."""
        real = next_d if (next_d is not None and self.real_ahead and hip.PROFILE is None) else None
        self._real_pending = real is not None
        self._real_key = self._batch_key(real) if real is not None else None
        if use_ahead and self._ahead_pending and hip.PROFILE is None:
            self._ahead_pending = False
            return self._run_step('g', batch, counter, use_ahead=True, real=real)
        self._ahead_pending = False
        return self._run_step('g', batch, counter, real=real)

    def _g_impl(self, batch, use_ahead=False, real=None):
        loss_g = self.g_gradients(batch, use_ahead, real)
        self._apply_g_launch()
        hip.mark('g/end')
        return loss_g

    def apply_g(self, counter=0):
        """optim_g.apply_gradients under control_dependencies(spectral-norm u assigns)."""
        self._adam_prepare(self.store.generator, 0, self.lr_g * self.decay(counter))
        self._apply_g_launch()

    def _apply_g_launch(self):
        self._allreduce_wait()
        self._adam_launch(self.store.generator, 0)
        if self.D.sn and self._sn_pending is not None:
            if hasattr(self.D, 'commit_u'):
                self.D.commit_u(self._sn_pending)
            else:
                self.store['discriminator/fully_connected/u'].copy_(self._sn_pending['u_new'])
            self._sn_pending = None
            hip.LAUNCHES += 1       # the assign may be a torch copy: the graph segment that holds it is not empty (_seg_end_graph)

    def g_gradients(self, batch, use_ahead=False, real=None):
        """loss_g and d loss_g / d generator variables; section all-reduces start as they finish."""
        hip.WGRAD_STREAM = self._wgrad_stream
        try:
            return self._g_gradients(batch, use_ahead, real)
        finally:
            hip.join_wgrad()
            hip.WGRAD_STREAM = None

    def _g_gradients(self, batch, use_ahead=False, real=None):
        B, s = self.bufs, self.store
        self._g_done = set()        # a step that was abandoned half way (failed capture) must not leave sections marked as sent
        N, _, H, W = batch['sketches'].shape
        if use_ahead:       # the forward pass of this batch was run during the discriminator step (_d_impl)
            xd_f, gctx = self._ahead
            sn = self.D.prepare_sn()
            img4 = B.get('img4', (N, H, W, 4), zero_on_alloc=True)
            hip.nchw_to_nhwc(batch['images'], img4, 0)
        elif self._aux_stream is not None and hip.PROFILE is None:
            # the spectral-norm power iterations and the target image's layout change do not depend on the generator
            self._aux_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._aux_stream):
                sn = self.D.prepare_sn()
                img4 = B.get('img4', (N, H, W, 4), zero_on_alloc=True)
                hip.nchw_to_nhwc(batch['images'], img4, 0)
            xd_f, gctx = self._pack_fake(batch)
            torch.cuda.current_stream().wait_stream(self._aux_stream)
        else:
            xd_f, gctx = self._pack_fake(batch)
            sn = self.D.prepare_sn()
            img4 = B.get('img4', (N, H, W, 4), zero_on_alloc=True)
            hip.nchw_to_nhwc(batch['images'], img4, 0)
        if real is not None:
            # the next discriminator step's real pass, on its own stream under this whole step (joined at its end); it starts
            # from the u this step assigns after its optimizer launch (u_new of the power iteration above)
            self._real_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._real_stream):
                hip.mark('g/next D(real) pass: first')
                self._real = self._d_real_pass(real, u=(sn['u_new'] if self.D.sn else None))
                hip.mark('g/next D(real) pass: last')
        hip.mark('g/start (D forward)')
        cf = self.D.forward(xd_f, sn, 'df')
        loss_g = self.loss[0:1]
        loss_g.zero_()
        rows = cf['disc'].shape[0] * cf['disc'].shape[1] * cf['disc'].shape[2]
        dl5_f = B.get('dl5_f', cf['disc'].shape, zero_on_alloc=True)
        hip.call('ssc_softplus_loss', cf['disc'], 4, rows, -1.0, 1.0 / rows, loss_g, dl5_f, 1.0 / rows)
        K = cf['logits'].shape[1]
        dlog_f = B.get('dlog_f', (N, K))
        hip.call('ssc_acgan_loss', cf['logits'], batch['class_id'], N, K, 0, 0.5, loss_g, dlog_f)
        hip.mark('g/D data gradient: first')
        dgen = self.D.backward(cf, dl5_f, dlog_f, sn, False, True, accumulate=False)
        hip.mark('g/G backward: first')
        dpre = B.get('dpre', (N, H, W, 4))
        hip.call('ssc_gen_output_grad', xd_f.view(-1)[3:], 8, img4, 4, dgen, 4, N * H * W, 100.0, loss_g, dpre)
        sc = s.generator
        if self.block_type == 'Pix2Pix':
            side = self._aux_stream if hip.PROFILE is None else None
            self.G.backward(gctx, dpre, on_section=lambda name: self._section_done(sc, name), side_stream=side)
        else:
            self.G.backward(gctx, dpre, on_section=lambda name: self._section_done(sc, name))
        if real is not None:
            torch.cuda.current_stream().wait_stream(self._real_stream)
        self._sn_pending = sn if self.D.sn else None
        self.last = {'gctx': gctx, 'cf': cf}
        return loss_g

    def _section_done(self, sc, name):
        s = self.store
        hip.join_wgrad()        # the section's filter gradients must have landed before anything reads them
        if name == 'decoders':      # the noise head's regulariser lives in this section
            hip.call('ssc_l2_reg', s['generator/fully_connected/weights'],
                     s['generator/fully_connected/weights'].numel(), 1e-6, self.loss[0:1],
                     s.grad('generator/fully_connected/weights'))
        lo, hi = self._g_sections[name]
        if self._g_merge:
            # fewer, larger exchanges (SSC_G_SECTIONS=2 | 1): every exchange ends a graph segment, and the side chains of the
            # backward pass (held filter gradients, the caption branch) cannot run across a segment end.  2: decoders + text as one
            # exchange when both are final, all encoders at the end; 1: one exchange at the end.
            self._g_done.add(name)
            sec = self._g_sections
            if self._g_merge == 2 and {'decoders', 'text'} <= self._g_done and 'dt' not in self._g_done:
                self._g_done.add('dt')
                self._allreduce_async(sc.grad, sec['text'][0], sec['decoders'][1])
            if name == 'encoders':
                hi = sec['text'][0] if self._g_merge == 2 else sec['decoders'][1]
                self._allreduce_async(sc.grad, 0, hi)
                self._g_done = set()
            return
        self._allreduce_async(sc.grad, lo, hi)

    def train_iteration(self, batch_d, batch_g, counter=0, next_batch_d=None):
        """D-step then G-step on independent batches (main_procedure.py:178-232).  Knowing both batches up front, the
        generator forward of the G-step is run inside the D-step (``run_ahead``; results identical: the generator's
        variables do not change in between).  Knowing the NEXT iteration's discriminator batch as well, its real pass is run
        inside this G-step (``real_ahead``; identical again: a G-step does not touch the discriminator's variables) -- pass
        that same batch as ``batch_d`` of the next call."""
        ld = self.d_step(batch_d, counter, ahead=batch_g, use_real=True)
        lg = self.g_step(batch_g, counter, use_ahead=True, next_d=next_batch_d)
        return lg, ld

    def _infer_inputs(self, kind, sketches, noise_vec, labels):
        """The static device tensors a replayed inference pass of this input shape reads."""
        ikey = ('infer_in', kind, tuple(sketches.shape), tuple(noise_vec.shape))
        st = self._static.get(ikey)
        if st is None:
            st = {'sk': torch.empty_like(sketches), 'nv': torch.empty_like(noise_vec),
                  'lb': None if labels is None else torch.empty_like(labels)}
            self._static[ikey] = st
        return st

    def infer_buffers(self, sketches, noise_vec, labels=None, kind='f32'):
        """(sketches, noise_vec, labels) as the tensors the replayed inference graphs read, allocated after the given ones and
        filled with them.  A serving loop that writes its next request into them and passes THEM to ``generate`` /
        ``generate_u8`` (kind 'u8') saves the per-call device copies (graphs replay fixed addresses)."""
        labels = None if (self.block_type != 'MRU' or labels is None) else labels.to(device='cuda', dtype=torch.int32).contiguous()
        st = self._infer_inputs(kind, sketches.contiguous(), noise_vec.contiguous(), labels)
        st['sk'].copy_(sketches)
        st['nv'].copy_(noise_vec)
        if labels is not None:
            st['lb'].copy_(labels)
        return st['sk'], st['nv'], st['lb']

    def _infer(self, kind, sketches, text, noise_vec, labels, thicken, clone=True):
        """The generator forward of ``generate`` ('f32': NCHW float in / out) and ``generate_u8`` ('u8': uint8 NHWC in /
        out, pre- and post-processing kernels included).  Like the training steps it runs eagerly the first time a
        shape is seen, is captured into a hipGraph the second time and replayed afterwards (a batch-16 forward is ~150
        launches of 5-100 us: launch-bound when issued one by one); inputs are copied into the graph's static tensors
        (unless they ARE those tensors: ``infer_buffers``) and the result is returned as a fresh tensor (clone=False: the
        graph's own output tensor, valid until the next call)."""
        if self.block_type == 'MRU' and labels is None:
            raise ValueError('the MRU generator is class-conditional: pass the class ids (image_data_class_id)')
        labels = None if self.block_type != 'MRU' else labels.to(device='cuda', dtype=torch.int32).contiguous()

        def body(sk, tx, nv, lb):
            xs = hip.sketch_preprocess_u8(sk, thicken) if kind == 'u8' else sk
            ctx = self.G.forward(xs, tx, lb, nv, 'g') if self.block_type == 'MRU' else self.G.forward(xs, tx, nv, 'g')
            return hip.image_postprocess_u8(ctx['out'], ctx['out_coff']) if kind == 'u8' else self.G.output_nchw(ctx)

        sketches, noise_vec = sketches.contiguous(), noise_vec.contiguous()
        if not self.use_graphs_infer or hip.PROFILE is not None:
            return body(sketches, text, noise_vec, labels)
        prep = text if isinstance(text, dict) or not self.G.lstm_hybrid else self.G.text.prepare(text, 'gi')
        S = prep['S'] if isinstance(prep, dict) else -1
        key = ('infer', kind, bool(thicken), bool(self.G.lstm_hybrid), tuple(sketches.shape), S)
        st = self._infer_inputs(kind, sketches, noise_vec, labels)
        for name, src in (('sk', sketches), ('nv', noise_vec), ('lb', labels)):
            if src is not None and src.data_ptr() != st[name].data_ptr():
                st[name].copy_(src)
        g = self._graphs.get(key)
        if g is None:
            if key not in self._seen:
                self._seen.add(key)
                return body(st['sk'], prep, st['nv'], st['lb'])
            try:
                g = hip.new_graph()
                with torch.cuda.graph(g, capture_error_mode='thread_local'):
                    self._static[key] = body(st['sk'], prep, st['nv'], st['lb'])
            except Exception as e:      # never lose a request to graph capture
                print('hipGraph capture of the inference pass failed (%r): continuing with eager launches' % (e,))
                self.use_graphs_infer = False
                torch.cuda.synchronize()
                return body(sketches, text, noise_vec, labels)
            self._graphs[key] = g
        hip.resplit_stale()         # weights replaced through torch (load_dict, a restore) since the planes were made
        g.replay()
        return self._static[key].clone() if clone else self._static[key]

    def generate_u8(self, sketch_u8, text, noise_vec, labels=None, thicken=False, clone=True):
        """Serving path without host arithmetic: uint8 sketches [N,H,W,3] on the device -> uint8 images [N,H,W,3].
        Pre-processing (x/255*2-1, optional thicken_drawings) and post-processing ((x+1)/2*255, truncating cast) of
        main_procedure.py:361-621 run as kernels; the network reads / writes NHWC directly."""
        return self._infer('u8', sketch_u8, text, noise_vec, labels, thicken, clone)

    def generate(self, sketches, text, noise_vec, labels=None, clone=True):
        """Inference path of build_single_graph (training=False): NCHW in, NCHW out.  clone=False: the replayed graph's own
        output tensor (overwritten by the next call); inputs from ``infer_buffers`` are read in place."""
        return self._infer('f32', sketches, text, noise_vec, labels, False, clone)


Pix2PixTrainer = GanTrainer     # the name the first round used
