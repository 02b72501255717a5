"""Background_Colorization generator (bg_colorization_main.py:302-420) behind the reference's function name.

Only ``create_residual_generator`` (forward) is built: BASELINE.json config 5, the 768x768 large-activation stress
case.  The BG module's discriminator, losses, feed_dict trainer and PNG writer are out of scope (SURVEY.md 8, row
A13 "next").  The reference builds the variables under tf.variable_scope('generator') (:582-586); they live in a
``ParamStore('BG')`` keyed by those TF names so a converted checkpoint loads with ``store.load_dict``.
"""
import numpy as np
import os

import torch

from . import hip
from .params import Buffers, ParamStore
from .residual import ResidualGenerator

_TOWERS = {}


def get_tower(image_size=768, vocab_size=18, ngf=64, seg_classes=3, seed=0, device='cuda'):
    key = (image_size, vocab_size, ngf, seg_classes)
    if key not in _TOWERS:
        if ngf != 64:
            raise NotImplementedError('ngf=%d: the parameter registry is laid out for the default ngf=64' % ngf)
        hip.lib()
        store = ParamStore('BG', vocab_size, image_size, device, seed)
        bufs = Buffers(device)
        _TOWERS[key] = (store, bufs, ResidualGenerator(store, bufs, 'bg', True, ngf, seg_classes))
    return _TOWERS[key]


def reset():
    _TOWERS.clear()


def create_residual_generator(generator_inputs, generator_outputs_channels, vocab_indices, ngf=64, vocab_size=18,
                              seg_classes=3, multi_residual=True):
    """generator_inputs NHWC [N,H,W,3] in [-1,1], vocab_indices int [N,T] ->
    (outputs NHWC [N,H,W,3] = tanh image, region_mask_logits NHWC [N,H,W,seg_classes])."""
    if not multi_residual or generator_outputs_channels != 3:
        raise NotImplementedError('only multi_residual=True, 3 output channels (the reference defaults, :805-807)')
    x = torch.as_tensor(np.asarray(generator_inputs) if not isinstance(generator_inputs, torch.Tensor)
                        else generator_inputs).to(device='cuda', dtype=torch.float32).contiguous()
    text = vocab_indices.cpu().numpy() if isinstance(vocab_indices, torch.Tensor) else np.asarray(vocab_indices)
    store, bufs, gen = get_tower(x.shape[1], vocab_size, ngf, seg_classes)
    ctx = gen.forward(x, text, None, 'bg')
    return ctx['image'], ctx['region_logits']


# ---------------------------------------------------------------------------------------------------------------
# training (create_model in train mode, bg_colorization_main.py:516-726)
# ---------------------------------------------------------------------------------------------------------------
class BGTrainer(object):
    """Generator + residual discriminator of the BG module with the reference's losses and optimizer:

        discrim_loss = mean(-(log(D(x, y) + eps) + log(1 - D(x, G(x)) + eps)))
        gen_loss     = gan_weight * mean(-log(D(x, G(x)) + eps)) + l1_weight * mean_{label != 0} |y - G(x)|
                       + seg_weight * mean CE(region logits, labels)
        Adam(lr_t, beta1 = 0.5, beta2 = 0.999) on both nets, lr_t = polynomial_decay(lr, step, 0.75 * max_steps,
        lr / 10, power 0.9).

    One ``train_step`` = one ``sess.run(model.train)`` (:898-901): a single forward pass, both gradient sets, both
    Adam applies.  The reference orders the generator's gradient ops after the discriminator update
    (control_dependencies, :648) without defining which discriminator weights they read; here both gradients are
    taken at the weights the forward pass used."""

    def __init__(self, image_size=768, vocab_size=18, ngf=64, ndf=64, seg_classes=3, lr=2e-4, max_steps=100000,
                 gan_weight=1.0, l1_weight=100.0, seg_weight=100.0, beta1=0.5, seed=0, device='cuda', use_graphs=True):
        if not torch.cuda.is_available():
            raise RuntimeError('BGTrainer needs an MI355X (HIP) device: there is no CPU fallback')
        if ngf != 64 or ndf != 64:
            raise NotImplementedError('the parameter registry is laid out for ngf = ndf = 64')
        from .residual import BGDiscriminator
        hip.lib()
        self.store = ParamStore('BG', vocab_size, image_size, device, seed)
        self.bufs = Buffers(device)
        self.G = ResidualGenerator(self.store, self.bufs, 'bg', True, ngf, seg_classes)
        self.D = BGDiscriminator(self.store, self.bufs, ndf)
        self.seg = seg_classes
        self.lr, self.max_steps = lr, max_steps
        self.w_gan, self.w_l1, self.w_seg = gan_weight, l1_weight, seg_weight
        self.beta1, self.beta2, self.eps = beta1, 0.999, 1e-8
        # [discrim_loss, gen_loss, gen_loss_GAN, gen_loss_L1, region_mask_loss] accumulated in double on the device
        self.losses = torch.zeros(5, dtype=torch.float64, device=device)
        for sc in (self.store.generator, self.store.discriminator):
            sc.adam_m = torch.zeros_like(sc.adam_v)
        self.global_step = 0
        # hipGraph replay of whole steps (as GanTrainer): ~1500 launches per step are otherwise issued one by one.  The
        # step sizes live in device memory so that one captured graph serves every step.
        self.use_graphs = bool(use_graphs)
        self.lr_dev = torch.zeros(2, dtype=torch.float32, device=device)
        self._static, self._graphs, self._seen = {}, {}, set()
        self._graph_gen = {}        # graph key -> hip.split_generation() at its capture
        # D(real) -- forward, its loss term, backward into the discriminator's gradient buffer -- does not depend on the generator:
        # it runs on a stream of its own beside the generator forward, in the CUs that pass's many small launches leave idle
        # (SSC_BG_OVERLAP_REAL=0: everything in line, as rounds 3-5)
        self._real_stream = torch.cuda.Stream() if os.environ.get('SSC_BG_OVERLAP_REAL', '1') == '1' else None
        # (filter gradients on a stream of their own beside the data-gradient chain, hip.WGRAD_STREAM: measured 24.3-24.8 vs 22.5 ms,
        # same bits -- not kept; profiles/NOTEBOOK_r06.md section 6)

    def learning_rate(self, step):
        decay_steps = int(round(self.max_steps * 0.75))
        s = min(step, decay_steps)
        return (self.lr - self.lr / 10.0) * (1.0 - s / decay_steps) ** 0.9 + self.lr / 10.0

    def _pack(self, name, inputs, second):
        N, H, W, _ = inputs.shape
        xd = self.bufs.get(name, (N, H, W, 8), zero_on_alloc=True)
        M = N * H * W
        hip.call('ssc_strided_copy', inputs, 3, xd, 8, M, 3, 0)
        hip.call('ssc_strided_copy', second, 3, xd.view(-1)[3:], 8, M, 3, 0)
        return xd

    def gradients(self, inputs, targets, text, labels_gt):
        """inputs / targets NHWC [N,H,W,3] in [-1,1], text int [N,T] (host), labels_gt int32 [N,H,W].
        Fills both flat gradient buffers and ``self.losses``; returns the generator context."""
        B = self.bufs
        inputs, targets = inputs.contiguous(), targets.contiguous()
        labels = labels_gt.to(device=inputs.device, dtype=torch.int32).contiguous()
        N, H, W, _ = inputs.shape
        M = N * H * W
        L = self.losses
        L.zero_()
        def real_pass():
            cr = self.D.forward(self._pack('xd_real', inputs, targets), 'dr')
            nz = cr['z'].numel()
            dz_r = B.get('dz_r', cr['z'].shape)
            hip.call('ssc_bg_gan_loss', cr['z'], nz, 0, 1.0 / nz, L[0:1], dz_r, 1.0 / nz)
            self.D.backward(cr, dz_r, True, False, accumulate=False)
            return cr

        side = self._real_stream if hip.PROFILE is None else None      # per-kernel timing runs everything in line
        if side is not None:
            main = torch.cuda.current_stream()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                cr = real_pass()
        gctx = self.G.forward(inputs, text, None, 'bg')
        image, logits = gctx['image'], gctx['region_logits']
        if side is None:
            cr = real_pass()
        cf = self.D.forward(self._pack('xd_fake', inputs, image), 'df')
        nz = cr['z'].numel()
        ws = hip.workspace()
        # ---- discriminator loss and gradients (the fake term adds to the loss word and the gradient buffer the real pass wrote)
        if side is not None:
            main.wait_stream(side)
        dz_f = B.get('dz_f', cf['z'].shape)
        hip.call('ssc_bg_gan_loss', cf['z'], nz, 1, 1.0 / nz, L[0:1], dz_f, 1.0 / nz)
        if side is not None:
            # the fake pair's backward into the discriminator's gradient buffer is off the critical path (nothing reads it before
            # the optimizer): on the side stream, beside the generator's loss terms and backward pass.  The two passes through
            # the same forward context then need gradient scratch of their own: the generator's pass takes the 'dg' set.
            side.wait_stream(main)
            with torch.cuda.stream(side):
                self.D.backward(cf, dz_f, True, False, accumulate=True)
            cfg = dict(cf, tag='dg', tape=[dict(rec, tag='dg') for rec in cf['tape']])
        else:
            self.D.backward(cf, dz_f, True, False, accumulate=True)
            cfg = cf
        # ---- generator loss and gradients
        dz_g = B.get('dz_g', cf['z'].shape)
        hip.call('ssc_bg_gan_loss', cf['z'], nz, 0, 1.0 / nz, L[2:3], dz_g, self.w_gan / nz)
        dgan = self.D.backward(cfg, dz_g, False, True, accumulate=False)
        count = B.get('l1_count', (1,))
        hip.call('ssc_count_nonzero_i32', labels, M, count, ws, ws.numel() * 4)
        dpre = B.get('dpre', (N, H, W, 4))
        hip.call('ssc_bg_output_grad', image, targets, labels, count, 1.0, dgan, L[3:4], dpre, M)
        if self.w_l1 != 1.0:        # the kernel folds the weight into the gradient: run it with the real weight
            L[3:4].zero_()
            hip.call('ssc_bg_output_grad', image, targets, labels, count, float(self.w_l1), dgan, L[3:4], dpre, M)
        dlog = B.get('dlog', (N, H, W, 4), zero_on_alloc=True)
        hip.call('ssc_seg_ce_loss', logits, self.seg, labels, M, float(self.w_seg), L[4:5], dlog, 4)
        self.G.backward(gctx, dpre, None, dlogits=dlog)
        if side is not None:
            main.wait_stream(side)      # the discriminator's gradient buffer is complete
        return gctx

    def loss_values(self):
        """(discrim_loss, gen_loss, gen_loss_GAN, gen_loss_L1, region_mask_loss) as Python floats."""
        d, _, gan, l1w, segw = [float(v) for v in self.losses.tolist()]
        l1 = l1w / self.w_l1 if self.w_l1 else 0.0
        seg = segw / self.w_seg if self.w_seg else 0.0
        return d, gan * self.w_gan + l1w + segw, gan, l1, seg

    def _adam_prepare(self):
        """Host part of both Adam applies: advance t, put lr_t = lr*sqrt(1-b2^t)/(1-b1^t) in device memory."""
        lr = self.learning_rate(self.global_step)
        for i, sc in enumerate((self.store.discriminator, self.store.generator)):
            sc.adam_t += 1
            t = sc.adam_t
            self.lr_dev[i:i + 1].fill_(float(lr * (1.0 - self.beta2 ** t) ** 0.5 / (1.0 - self.beta1 ** t)))
        self.global_step += 1

    def _adam_launch(self):
        for i, sc in enumerate((self.store.discriminator, self.store.generator)):
            hip.call('ssc_adam_tf', sc.flat, sc.grad, sc.adam_m, sc.adam_v, sc.numel, 0.0, self.lr_dev[i:i + 1],
                     self.beta1, self.beta2, self.eps, 1.0)
            hip.refresh_splits(sc.flat)     # the bf16 planes of this scope's filters follow the weights

    def apply_gradients(self):
        self._adam_prepare()
        self._adam_launch()

    def train_step(self, inputs, targets, text, labels_gt):
        """One ``sess.run(model.train)``.  The first call of a shape runs eagerly (it allocates), the second is captured
        into a hipGraph, later ones replay it on the inputs copied into the graph's static tensors."""
        if not self.use_graphs or hip.PROFILE is not None:
            gctx = self.gradients(inputs, targets, text, labels_gt)
            self.apply_gradients()
            return gctx
        skey = tuple(inputs.shape)
        st = self._static.get(skey)
        if st is None:
            st = {'inputs': torch.empty_like(inputs.contiguous()), 'targets': torch.empty_like(targets.contiguous()),
                  'labels': torch.empty(tuple(labels_gt.shape), dtype=torch.int32, device=inputs.device)}
            self._static[skey] = st
        st['inputs'].copy_(inputs)
        st['targets'].copy_(targets)
        st['labels'].copy_(labels_gt)
        prep = text if isinstance(text, dict) else self.G.text.prepare(text, 'bg')
        key = skey + (prep['S'],)
        self._adam_prepare()

        def impl():
            self._gctx = self.gradients(st['inputs'], st['targets'], prep, st['labels'])
            self._adam_launch()

        g = self._graphs.get(key)
        if g is None:
            if key not in self._seen:
                self._seen.add(key)
                impl()
                return self._gctx
            try:
                g = hip.new_graph()
                with torch.cuda.graph(g, capture_error_mode='thread_local'):
                    impl()
            except Exception as e:      # never lose a training run to graph capture
                print('hipGraph capture failed (%r): continuing with eager launches' % (e,))
                self.use_graphs = False
                torch.cuda.synchronize()
                impl()
                return self._gctx
            self._graphs[key] = g
            self._graph_gen[key] = hip.split_generation()
        hip.resplit_stale()         # weights replaced through torch since the planes were made (trainer.py: _run_step_inner)
        g.replay()
        for sc in (self.store.discriminator, self.store.generator):
            hip.refresh_new_splits(sc.flat, self._graph_gen[key])
        return self._gctx
