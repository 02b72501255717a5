"""Graph assembly API of obj_lib/graph_single.py over the eager HIP tower.

  build_single_graph(...)       reference :221-314   one tower: [gen, images, sketches] or
                                                      (loss_g, loss_d, grad_g, grad_d)
  build_multi_tower_graph(...)  reference :107-218   -> (opt_g, opt_d, loss_g, loss_d, summaries)

The reference returns symbolic tensors/ops to be fetched with ``sess.run``; the objects returned here
are fetched with ``Session().run([...])`` with the same grouping semantics: fetching ``opt_d`` runs one
discriminator update on a freshly dequeued batch, fetching ``opt_g`` one generator update.
Multi-GPU: the reference loops towers inside one process; here every process is one tower
(torch.distributed world_size == num_gpu) and average_gradients (:33-68) is an RCCL all-reduce.
"""
import numpy as np
import torch

from . import models_collection as models
from .config import Config
from .input_pipeline import get_num_classes, split_inputs


def _value(x):
    return x() if callable(x) else x


def _dev(x, dtype=torch.float32):
    if not isinstance(x, torch.Tensor):
        x = torch.as_tensor(np.asarray(x))
    if not x.is_cuda:       # through pinned memory, not waited for (a pageable copy returns behind everything queued on the stream)
        x = x.to(dtype).pin_memory().to('cuda', non_blocking=True)
    return x.to(device='cuda', dtype=dtype).contiguous()


def _batch(images, sketches, images_d, cls, cls_d, text, noise_vec=None):
    text = _value(text)
    text = text.cpu().numpy() if isinstance(text, torch.Tensor) else np.asarray(text)
    sk = _dev(_value(sketches))
    n = sk.shape[0]
    b = {'sketches': sk, 'text': text.astype(np.int32),
         'noise_vec': _dev(noise_vec) if noise_vec is not None else torch.randn(n, 256, device='cuda')}
    if images is not None:
        b['images'] = _dev(_value(images))
    if images_d is not None:
        b['images_d'] = _dev(_value(images_d))
    if cls is not None:
        b['class_id'] = _dev(_value(cls), torch.int32)
    if cls_d is not None:
        b['class_id_d'] = _dev(_value(cls_d), torch.int32)
    return b


def get_optimizer(optimizer_name, **kwargs):
    """graph_single.py:584-593: the optimizer family and its fixed hyper-parameters."""
    table = {'rmsprop': {'name': 'RMSProp', 'decay': 0.9, 'momentum': 0.0, 'epsilon': 1e-10},
             'adam': {'name': 'Adam', 'beta1': 0.0, 'beta2': 0.9},
             'adadelta': {'name': 'AdaDelta', 'rho': 0.95, 'epsilon': 1e-8},
             'adagrad': {'name': 'AdaGrad', 'initial_accumulator_value': 0.1}}
    if optimizer_name.lower() not in table:
        raise ValueError('unknown optimizer %r' % optimizer_name)
    return table[optimizer_name.lower()]


def build_single_graph(images, sketches, images_d, image_data_class_id, image_data_class_id_d,
                       text_vocab_indiceses, batch_size, training, LSTM_hybrid, vocab_size, ld=10,
                       data_format='NCHW', distance_map=True, optim_g=None, optim_d=None, block_type='MRU',
                       noise_vec=None):
    """One tower.  training=False: [image_gens, images, sketches] (reference :265-266).
    training=True: (loss_g, loss_d, grad_g, grad_d) with grad_* = list of (gradient, variable name)."""
    assert block_type in ['MRU', 'Pix2Pix', 'Residual']
    models.set_param(data_format=data_format)
    sk = _dev(_value(sketches))
    assert sk.shape[0] == batch_size, 'batch_size is baked into the reference graph (models_collection.py:162)'
    tr = models.get_trainer(block_type, vocab_size, sk.shape[2])
    tr.G.lstm_hybrid = bool(LSTM_hybrid)
    if not training:
        b = _batch(None, sk, None, image_data_class_id, None, text_vocab_indiceses, noise_vec)
        if block_type == 'MRU':     # class-conditional norms (models_collection.py:80-82, 270-272)
            gen = tr.generate(b['sketches'], b['text'], b['noise_vec'], labels=b['class_id'])
        else:
            gen = tr.generate(b['sketches'], b['text'], b['noise_vec'])
        return [gen, _value(images), sk]
    b = _batch(images, sk, images_d, image_data_class_id, image_data_class_id_d, text_vocab_indiceses, noise_vec)
    loss_d = float(tr.d_gradients(b))
    grad_d = [(g.clone(), n) for n, g in tr.store.discriminator.g.items()]
    loss_g = float(tr.g_gradients(b))
    grad_g = [(g.clone(), n) for n, g in tr.store.generator.g.items()]
    return loss_g, loss_d, grad_g, grad_d


class Fetch(object):
    def __init__(self, graph, kind):
        self.graph, self.kind = graph, kind


class LazyLoss(object):
    """A fetched loss whose value is read when somebody looks at it (``Session.run(..., lazy=True)``): the device value is
    copied to pinned host memory behind the step that produced it, and ``float()`` / ``np.asarray()`` wait for that copy
    only -- not for whatever was launched after it.  The training loop launches the next step first and tests the previous
    one for NaN while it runs (main_procedure.train); a tf.Session.run returns when the step is done, which on a device
    that replays a step in 6 ms leaves it idle for as long as the host needs to prepare the next one.  Reading the value
    also finishes the hand-off check (hip.check_sk) the fetch started."""
    _ring = {'host': None, 'i': 0}

    def __init__(self, dev_value, sk_handle, where):
        r = LazyLoss._ring
        if r['host'] is None:
            r['host'] = torch.zeros(64, dtype=torch.float64).pin_memory()
        r['i'] = (r['i'] + 1) % 64
        self._slot = r['host'][r['i']:r['i'] + 1]
        self._slot.copy_(dev_value.detach().reshape(-1)[:1].to(torch.float64), non_blocking=True)
        self._ev = torch.cuda.Event()
        self._ev.record()
        self._sk, self._where, self._val = sk_handle, where, None

    def value(self):
        if self._val is None:
            self._ev.synchronize()
            self._val = np.float32(float(self._slot[0]))
            from .. import hip
            sk, self._sk = self._sk, None
            hip.check_sk_end(sk, self._where)
        return self._val

    def __float__(self):
        return float(self.value())

    def __array__(self, dtype=None, copy=None):
        return np.asarray(self.value(), dtype=dtype)

    def __repr__(self):
        return repr(self.value())


class Counter(object):
    """tf.Variable(int32) + assign_add used for the lr schedule (main_procedure.py:105-106)."""

    def __init__(self, value=0):
        self.value = int(value)

    def assign(self, v):
        self.value = int(v)


class TowerGraph(object):
    def __init__(self, trainer, inputs, counter, rank, world, batch_size, batch_portion, pg=None):
        self.tr, self.inputs, self.counter = trainer, inputs, counter
        self.rank, self.world, self.batch_size, self.batch_portion = rank, world, batch_size, batch_portion
        self.pg = pg
        self.last = {'loss_g': float('nan'), 'loss_d': float('nan')}
        # inputs marked ``per_tower`` already hold this tower's slice (a queue per process that dequeues batch_size
        # examples: no rank reads and decodes the other towers' share); anything else is the global batch and is cut
        # by split_inputs as in the reference (graph_single.py:128-135)
        self.per_tower = all(getattr(x, 'per_tower', False) for x in inputs)
        if self.per_tower and world > 1 and len(set(int(p) for p in batch_portion[:world])) > 1:
            # split_inputs gives tower i batch_size * batch_portion[i] samples; per-tower queues all dequeue batch_size
            raise ValueError('per-tower input queues dequeue batch_size examples on every rank: a non-uniform '
                             'batch_portion %s needs global-batch inputs (split_inputs)' % (list(batch_portion),))

    def _dequeue(self):
        vals = [_value(x) for x in self.inputs]
        if self.world > 1 and not self.per_tower:      # split_inputs: this process is tower `rank`
            vals = [split_inputs(v, self.batch_size, self.batch_portion, self.world)[self.rank] for v in vals]
        images, sketches, images_d, cls, cls_d, text = vals
        return _batch(images, sketches, images_d, cls, cls_d, text)

    def _tower_mean(self, loss):
        """The loss every rank reports and tests for NaN: the mean over towers (sum all-reduce, so one tower's NaN is
        every rank's NaN and all ranks take the same restart branch -- ranks deciding on their local loss would leave
        the others waiting in the next gradient all-reduce).  The reference fetches the last tower's loss
        (graph_single.py:144-173, the loop variable) and sees another tower's NaN one iteration later."""
        if self.world == 1:
            return loss
        import torch.distributed as dist
        t = loss.detach().clone()
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.pg)
        return t / self.world

    def run(self, fetches, g_follows=False, d_follows=False, lazy=False):
        """g_follows (with opt_d): the next run is opt_g -- its batch is dequeued now (same order as the reference's
        queue: D batch, then G batch) so that the trainer can run its generator forward inside the D-step.
        d_follows (with opt_g): the next run is opt_d -- its batch is dequeued now (again the queue order is unchanged:
        ..., G batch, next D batch) so that the trainer can run that step's real pass inside this G-step.
        lazy: losses come back as LazyLoss -- the call returns when the step is LAUNCHED, the value (and the hand-off check)
        is read when it is looked at."""
        kinds = [f.kind for f in fetches]
        c = self.counter.value if isinstance(self.counter, Counter) else int(_value(self.counter))
        if 'opt_d' in kinds:
            d_batch, self._d_next = getattr(self, '_d_next', None), None
            use_real = d_batch is not None
            if d_batch is None:
                d_batch = self._dequeue()
            self._g_next = self._dequeue() if (g_follows and getattr(self.tr, 'run_ahead', False)) else None
            self.last['loss_d'] = self._tower_mean(self.tr.d_step(d_batch, c, ahead=self._g_next, use_real=use_real))
        if 'opt_g' in kinds:
            g_next, self._g_next = getattr(self, '_g_next', None), None
            g_batch = g_next if g_next is not None else self._dequeue()
            self._d_next = self._dequeue() if (d_follows and getattr(self.tr, 'real_ahead', False)) else None
            self.last['loss_g'] = self._tower_mean(self.tr.g_step(g_batch, c, use_ahead=g_next is not None,
                                                                   next_d=self._d_next))
        out = []
        sk_handle = None
        if 'loss_g' in kinds or 'loss_d' in kinds:
            # the loss is read back here anyway: also read the conv launches' hand-off timeout words, so that a launch that
            # stored a partial sum fails the run instead of training on (ssc_conv_desc.sk_flags, hip.check_sk)
            from .. import hip
            if lazy:
                sk_handle = hip.check_sk_begin()
            else:
                hip.check_sk('Session.run')
        for k in kinds:
            if k in ('loss_g', 'loss_d') and lazy and isinstance(self.last[k], torch.Tensor):
                out.append(LazyLoss(self.last[k], sk_handle, 'Session.run'))
                sk_handle = None        # one reader finishes the check
            elif k in ('loss_g', 'loss_d'):
                out.append(np.float32(float(self.last[k])))
            elif k == 'counter':
                out.append(c)
            elif k == 'counter_add':
                self.counter.value += 1
                out.append(self.counter.value)
            elif k == 'summaries':
                out.append({'total_loss/g': float(self.last['loss_g']), 'total_loss/d': float(self.last['loss_d']),
                            'learning_rate_g': self.tr.lr_g * self.tr.decay(c)})
            else:
                out.append(None)
        return out


class Session(object):
    """Minimal stand-in for tf.Session.run over Fetch objects."""

    def run(self, fetches, **kw):
        single = not isinstance(fetches, (list, tuple))
        fl = [fetches] if single else list(fetches)
        res = fl[0].graph.run(fl, **kw)
        return res[0] if single else res


def build_multi_tower_graph(images, sketches, images_d, image_paired_class_ids, image_paired_class_ids_d,
                            text_vocab_indiceses, LSTM_hybrid, vocab_size, batch_size, num_gpu, batch_portion,
                            training, learning_rates, counter, max_iter_step, ld=10, data_format='NCHW',
                            distance_map=True, optimizer='Adam', block_type='MRU'):
    """Inputs are tensors/arrays or zero-argument callables (the reference passes queue outputs).
    Returns (opt_g, opt_d, loss_g, loss_d, summaries) to be fetched through Session.run."""
    models.set_param(data_format=data_format)
    get_optimizer(optimizer)
    # distance_map only changes what the input queue feeds (input_pipeline.py:86-96); the graph ignores it
    pg, rank, world = None, 0, 1
    if num_gpu > 1:
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() == num_gpu):
            raise RuntimeError('num_gpu=%d needs one process per GPU (the reference looped towers in one process): run it '
                               'through obj_colorization_main.py -gpu %d (which starts its own ranks, dist_utils.launch_towers) '
                               'or `python -m torch.distributed.run --nproc-per-node %d ...`' % (num_gpu, num_gpu, num_gpu))
        pg, rank, world = dist.group.WORLD, dist.get_rank(), num_gpu
    # image size: from the input's ``img_size`` attribute when it has one (a queue output: nothing may be dequeued
    # here, or the first training batch would pair batch-0 sketches with batch-1 images), else from the value
    probe = None
    img = getattr(sketches, 'img_size', None)
    if img is None:
        probe = _value(sketches)
        img = probe.shape[2]
    # the training procedure's steps are captured into hipGraphs and replayed (as bench.py's are): an eager step is ~380 launches
    # of ~30 us of interpreter time each -- as long as the device needs for the step at batch 32, and the host has the input
    # queue to serve as well.  SSC_TRAIN_GRAPHS=0: every launch issued by the interpreter
    import os
    tr = models.get_trainer(block_type, vocab_size, img, process_group=pg, optimizer=optimizer,
                            use_graphs=os.environ.get('SSC_TRAIN_GRAPHS', '1') == '1')
    tr.G.lstm_hybrid = bool(LSTM_hybrid)
    tr.lr_g, tr.lr_d = learning_rates['generator'], learning_rates['discriminator']
    tr.max_iter_step = max_iter_step
    if callable(sketches) and probe is not None:      # the probed value is the first dequeue
        first = {'v': probe}
        orig = sketches

        def sketches():
            return first.pop('v') if 'v' in first else orig()
        sketches.per_tower = getattr(orig, 'per_tower', False)
    g = TowerGraph(tr, [images, sketches, images_d, image_paired_class_ids, image_paired_class_ids_d,
                        text_vocab_indiceses], counter, rank, world, batch_size, list(batch_portion), pg)
    return Fetch(g, 'opt_g'), Fetch(g, 'opt_d'), Fetch(g, 'loss_g'), Fetch(g, 'loss_d'), Fetch(g, 'summaries')
