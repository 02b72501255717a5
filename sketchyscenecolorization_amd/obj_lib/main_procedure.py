"""Procedures of obj_lib/main_procedure.py: train (:62-242), validation (:245-358), test (:361-492),
inference (:495-621) -- same names, same Config-driven behaviour, same output files -- over the HIP tower.

Differences that are forced by the environment and kept observable-equivalent:
  * TFRecord queues are out of scope: ``train`` draws batches from
    ``Config.data_source`` ('synthetic' by default: sketchyscenecolorization_amd.synthetic);
  * checkpoints are ``torch.save`` dicts keyed by the TF variable names, written under the reference's
    file names (``snapshot/model_<i>.ckpt-<i>`` + a ``checkpoint`` index like tf.train.Saver);
  * images are written with PIL (cv2 is absent); the reference's RGB->BGR flip + cv2.imwrite
    (main_procedure.py:609-621) yields the same RGB file content.
"""
import json
import os
from time import time

import numpy as np
import torch

from ..data_processing.text_processing import load_vocab_dict_from_file, preprocess_sentence
from . import models_collection as models
from .config import Config
from .graph_single import Counter, Session, build_multi_tower_graph, build_single_graph
from .input_pipeline import resize_and_padding_mask_image, thicken_drawings

PKG_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CATEGORIES = ['bench', 'bird', 'bus', 'butterfly', 'car', 'cat', 'chair', 'chicken', 'cloud', 'cow', 'dog', 'duck',
              'grass', 'horse', 'house', 'moon', 'person', 'pig', 'rabbit', 'road', 'sheep', 'star', 'sun', 'tree',
              'truck']      # sorted(os.listdir('data/captions')) of the reference dataset
SIZE = {True: (64, 64), False: (192, 192)}


# ----------------------------------------------------------------------------- checkpoints (tf.train.Saver look-alike)
def save_checkpoint(store, ckpt_dir, name, global_step):
    from .. import hip
    hip.check_sk('save_checkpoint')         # never snapshot weights behind a launch that reported a partial sum
    os.makedirs(ckpt_dir, exist_ok=True)
    path = os.path.join(ckpt_dir, '%s-%d' % (name, global_step))
    torch.save(store.state_dict(), path)
    with open(os.path.join(ckpt_dir, 'checkpoint'), 'w') as f:
        f.write('model_checkpoint_path: "%s"\n' % os.path.basename(path))
    return path


def latest_checkpoint(ckpt_dir):
    idx = os.path.join(ckpt_dir, 'checkpoint')
    if not os.path.exists(idx):
        return None
    with open(idx) as f:
        line = f.readline()
    name = line.split('"')[1]
    path = os.path.join(ckpt_dir, os.path.basename(name))
    return path if (os.path.exists(path) or os.path.exists(path + '.index')) else None


def restore_checkpoint(store, path):
    """Our own snapshots (torch files) or a TensorFlow V2 checkpoint prefix written by tf.train.Saver -- e.g. the
    authors' released models dropped into outputs/<stamp>/snapshot: variables are matched by their TF names, the
    Adam second-moment slots ('<var>/Adam_1') and step counts ('beta2_power') are taken over when present."""
    from .. import tf_checkpoint
    if not tf_checkpoint.is_tf_checkpoint(path):
        store.load_state_dict(torch.load(path, map_location='cpu'))
        return
    tensors = tf_checkpoint.read_checkpoint(path)
    store.load_dict(tensors)
    missing = [n for n in store.names() if n not in tensors]
    if missing:
        print('restore_checkpoint: %d variables not in %s (kept at their initial values), e.g. %s'
              % (len(missing), path, missing[:3]))
    for sc in (store.generator, store.discriminator):
        for n, (off, k, _shape) in sc.offsets.items():
            slot = tensors.get(n + '/Adam_1')
            if slot is not None and slot.size == k:
                sc.adam_v[off:off + k].copy_(torch.from_numpy(slot.reshape(-1)))
    # Adam step counts.  beta2_power = 0.9 ** t of each optimizer (generator built first) recovers t only while the
    # float32 power has not underflowed (t < ~1000); for any longer-trained checkpoint the global step in the file name
    # ('model_<i>.ckpt-<i>': one D and one G apply per iteration) stands in, so that the bias correction
    # sqrt(1 - 0.9^t) is ~1 as it was when the checkpoint was written instead of restarting at 0.316.
    step = None
    tail = os.path.basename(path).rsplit('-', 1)
    if len(tail) == 2 and tail[1].isdigit():
        step = int(tail[1]) + 1
    seen = set()
    for key, val in tensors.items():
        leaf = key.split('/')[-1]
        if leaf.startswith('beta2_power'):
            sc = store.discriminator if leaf.endswith('_1') else store.generator
            if 0.0 < float(val) < 1.0:
                sc.adam_t = int(round(np.log(float(val)) / np.log(0.9)))
                seen.add(id(sc))
    for sc in (store.generator, store.discriminator):
        if id(sc) not in seen and any((n + '/Adam_1') in tensors for n in sc.offsets):
            sc.adam_t = step if step is not None else 100000


def print_parameter_count(store, verbose=False):
    for scope in ('generator', 'discriminator'):
        print(scope)
        print('total_parameters', store.parameter_count(scope))


# ----------------------------------------------------------------------------- data
class SyntheticQueue(object):
    """Stands in for build_input_queue_paired (input_pipeline.py:131-154): a pool of seeded batches."""

    def __init__(self, batch_size, img, vocab_size, seed, pool=4):
        from ..synthetic import synthetic_batch
        self.pool = [synthetic_batch(batch_size, seed + i, img, vocab_size) for i in range(pool)]
        self.i = 0
        self.cur = self.pool[0]
        self.img = img

    def advance(self):
        self.cur = self.pool[self.i % len(self.pool)]
        self.i += 1

    def field(self, name, advance=False):
        def f():
            if advance:
                self.advance()
            return self.cur[name]
        f.img_size, f.per_tower = self.img, True
        return f


class RecordQueue(object):
    """The same interface over the reference's TFRecords: queue 1 feeds (images, sketches, class_id, text), an
    independently shuffled queue 2 the discriminator's real images and labels (main_procedure.py:109-122)."""

    def __init__(self, batch_size, small, which, data_base_dir='data', seed=None):
        from .input_pipeline import PairedQueue
        # prefetch: the host half of the next batch (read, CRC, parse, pinned staging) on a thread of its own, in dequeue order
        self.q = PairedQueue('train', batch_size, Config.data_format, Config.distance_map != 0, small,
                             data_base_dir=data_base_dir, seed=seed,
                             prefetch=os.environ.get('SSC_RECORD_PREFETCH', '1') == '1')
        self.which = which
        self.cur = None
        self.img = SIZE[bool(small)][0]

    def advance(self):
        images, sketches, class_id, text = self.q.dequeue()
        # device-decoded batches stay put; host arrays go through pinned memory, not waited for (a copy from pageable memory
        # returns when it has happened: behind the training step that is running)
        dev = lambda a: a if torch.is_tensor(a) else torch.from_numpy(a).pin_memory().to('cuda', non_blocking=True)
        if self.which == 1:
            self.cur = {'images': dev(images), 'sketches': dev(sketches), 'class_id': dev(class_id), 'text': text}
        else:
            self.cur = {'images_d': dev(images), 'class_id_d': dev(class_id)}

    def field(self, name, advance=False):
        def f():
            if advance or self.cur is None:
                self.advance()
            return self.cur[name]
        f.img_size, f.per_tower = self.img, True
        return f


def _write_png(path, arr_uint8):
    from PIL import Image
    from .. import hip
    hip.check_sk('before writing %s' % os.path.basename(path))
    Image.fromarray(arr_uint8).save(path)


def _postprocess(nchw):
    """NCHW float [-1,1] -> NHWC uint8 with the reference's truncating cast (main_procedure.py:601-610)."""
    x = np.transpose(np.asarray(nchw.detach().cpu() if isinstance(nchw, torch.Tensor) else nchw), (0, 2, 3, 1))
    return (((x + 1) / 2.) * 255).astype(np.uint8)


# ----------------------------------------------------------------------------- train
def _settle_first(pending, n):
    """Read the first n pending losses (the steps launched before the one just issued): -1 at the first NaN."""
    for _ in range(n):
        who, val = pending.pop(0)
        if np.isnan(np.sum(val)):
            del pending[:]
            print("NaN occurred during training %s" % who)
            return -1
    return 0


def train(**kwargs):
    status = 0
    batch_size = Config.batch_size
    max_iter_step = Config.max_iter_step
    Diters = Config.disc_iterations
    num_gpu = Config.num_gpu
    ckpt_dir = Config.ckpt_dir
    log_dir = Config.log_dir
    small = Config.small_img != 0
    LSTM_hybrid = Config.LSTM_hybrid != 0
    distance_map = Config.distance_map != 0
    summary_write_freq = Config.summary_write_freq
    save_model_freq = Config.save_model_freq
    count_left_time_freq = Config.count_left_time_freq
    batch_portion = np.array([1, 1, 1, 1] + [1] * max(0, num_gpu - 4), dtype=np.int32)
    iter_from = kwargs['iter_from']
    img = SIZE[small][0]

    if num_gpu > 1:
        from ..dist_utils import init_distributed
        dist = init_distributed()
    rank = int(os.environ.get('RANK', 0))

    models.reset_default_graph()
    print('Iteration starts from: %d' % iter_from)
    counter = Counter(iter_from)

    # two INDEPENDENT queues, as in the reference (main_procedure.py:109-122): the discriminator's
    # "real" images are not paired with the sketches it sees
    # Every process is one tower and owns its queues: it dequeues batch_size examples per step (its share of the
    # reference's batch_size * num_gpu dequeue, input_pipeline.py:143-148 + split_inputs), from its own shuffle.
    if os.path.isdir(os.path.join('data', 'tfrecord', 'train')):     # the reference's dataset location (:109-122)
        q1 = RecordQueue(batch_size, small, 1, seed=(None if num_gpu == 1 else 7919 * rank + 1))
        q2 = RecordQueue(batch_size, small, 2, seed=(None if num_gpu == 1 else 7919 * rank + 2))
    else:
        print('data/tfrecord/train not found: training on seeded synthetic batches')
        q1 = SyntheticQueue(batch_size, img, Config.vocab_size, seed=1234 + 1000 * rank)
        q2 = SyntheticQueue(batch_size, img, Config.vocab_size, seed=998244 + 1000 * rank)
    opt_g, opt_d, loss_g, loss_d, merged_all = build_multi_tower_graph(
        q1.field('images', advance=True), q1.field('sketches'), q2.field('images_d', advance=True),
        q1.field('class_id'), q2.field('class_id_d'), q1.field('text'),
        LSTM_hybrid=LSTM_hybrid, vocab_size=Config.vocab_size, batch_size=batch_size, num_gpu=num_gpu,
        batch_portion=batch_portion, training=True,
        learning_rates={"generator": Config.lr_G, "discriminator": Config.lr_D},
        counter=counter, max_iter_step=max_iter_step, ld=Config.ld, data_format=Config.data_format,
        distance_map=distance_map, optimizer=Config.optimizer, block_type=Config.block_type)
    tower = opt_g.graph
    store = tower.tr.store
    sess = Session()
    if iter_from > 0:
        path = latest_checkpoint(ckpt_dir)
        print('Restore:', path)
        restore_checkpoint(store, path)
    print_parameter_count(store)
    counter.assign(iter_from)
    log_f = open(os.path.join(log_dir, 'scalars.jsonl'), 'a') if rank == 0 else None
    prev_time = float("-inf")
    fetch_counter, fetch_add = type(opt_g)(tower, 'counter'), type(opt_g)(tower, 'counter_add')

    # The losses of an iteration are READ one launch later (graph_single.LazyLoss): sess.run returns when a step is launched,
    # the next step is prepared and launched, and only then is the previous one tested for NaN -- while the device works.
    # (Reading every loss where the reference does leaves a 6 ms step waiting for the host twice per iteration: 14.05 vs
    # 12.9 ms per iteration at batch 32, scripts/cli_train_rate.sh.)  What the reference's order guarantees is kept: an
    # iteration whose loss is NaN ends train() with -1 before any LATER snapshot or scalar line is written (a step launched
    # in between is discarded with the process state: the caller restarts from the last snapshot); iterations that write a
    # scalar line or a snapshot, and the last one, read their losses before they do.  SSC_CLI_LAZY_LOSS=0: every loss read
    # where it is fetched.
    lazy = os.environ.get('SSC_CLI_LAZY_LOSS', '1') == '1'
    pending = []            # (which step, loss) of launches whose losses nobody has looked at yet

    for i in range(iter_from, max_iter_step):
        if i % count_left_time_freq == 0:
            curr_time = time()
            elapsed = curr_time - prev_time
            print("Now at iteration %d. Elapsed time: %.5fs. Average time: %.5fs/iter" % (i, elapsed, elapsed / 100.))
            if elapsed != float("inf"):
                left_sec = (max_iter_step - i) * (elapsed / 100.)
                d_ = int(left_sec / 86400)
                h_ = int((left_sec - 86400 * d_) / 3600)
                m_ = int((left_sec - 86400 * d_ - 3600 * h_) / 60)
                print("Left time:%dd %dh %dm" % (d_, h_, m_))
            prev_time = curr_time
        for j in range(Diters):
            # g_follows: the trainer may run the G-step's generator forward inside the last D-step (trainer.run_ahead)
            _, loss_d_out = sess.run([opt_d, loss_d], g_follows=(j == Diters - 1), lazy=lazy)
            earlier = len(pending)
            pending.append(('D', loss_d_out))
            # lazy: what was launched BEFORE this step is read while this step runs; else this step's loss here and now
            if _settle_first(pending, earlier if lazy else earlier + 1) == -1:
                return -1
        # d_follows: the trainer may run the next iteration's real D pass inside this G-step (trainer.real_ahead; on by
        # default for the Pix2Pix pair on one GPU).  That dequeues the next D batch one sess.run early -- the queue ORDER is unchanged (..., G batch, next D
        # batch), but a snapshot written after this step would see the queue one batch ahead of the reference, so iterations
        # that write a snapshot do not prefetch (the scalar summary reads no queue).
        snapshot_iter = i % save_model_freq == save_model_freq - 1
        _, loss_g_out, counter_out, _ = sess.run([opt_g, loss_g, fetch_counter, fetch_add],
                                                 d_follows=(Diters >= 1 and i + 1 < max_iter_step and not snapshot_iter),
                                                 lazy=lazy)
        pending.append(('G', loss_g_out))
        writes = (log_f is not None and i % summary_write_freq == 0) or snapshot_iter or i + 1 == max_iter_step
        if not lazy or writes or num_gpu > 1:       # (many towers: one read per iteration, behind the G-step's launch)
            if _settle_first(pending, len(pending)) == -1:
                return -1
        if log_f is not None and i % summary_write_freq == 0:
            summ = sess.run(merged_all)
            summ['step'] = i
            log_f.write(json.dumps(summ) + '\n')
            log_f.flush()
        if i % save_model_freq == save_model_freq - 1:
            if rank == 0:
                save_checkpoint(store, ckpt_dir, 'model_{}.ckpt'.format(i), global_step=i)
                print('Save model_{}.ckpt'.format(i))
            if num_gpu > 1:     # a restart on any rank must find the snapshot rank 0 has just written
                import torch.distributed as dist
                dist.barrier()
    return status


# ----------------------------------------------------------------------------- inference / test / validation
def _load_sketch(path, img_dim, category):
    from PIL import Image
    sketch_image = Image.open(path).convert("RGB")
    if sketch_image.width != img_dim[0] or sketch_image.height != img_dim[1]:
        margin_size = 0 if category in ['road'] else 10
        sketch = resize_and_padding_mask_image(sketch_image, img_dim[0], margin_size=margin_size).astype(np.float32)
    else:
        sketch = np.array(sketch_image, dtype=np.float32)
    return sketch


def _normalise(sketch_hwc):
    x = sketch_hwc / 255. * 2. - 1
    return np.transpose(np.expand_dims(x, axis=0), [0, 3, 1, 2]).astype(np.float32)


def _categories():
    captions_base_dir = os.path.join('data', 'captions')
    if os.path.isdir(captions_base_dir):
        c = os.listdir(captions_base_dir)
        c.sort()
        return c
    return list(CATEGORIES)


def _load_vocab():
    """data/vocab.txt of the working directory (the reference's location) or the built-in default list."""
    if os.path.exists('data/vocab.txt'):
        return load_vocab_dict_from_file('data/vocab.txt')
    from ..data_processing.default_vocab import default_vocab_dict
    return default_vocab_dict()


def inference(img_name, instruction):
    """One sketch + one caption -> <name>_output.png / <name>_input.png (main_procedure.py:495-621)."""
    wild_data_base_dir = 'examples'
    wild_cate = img_name[:img_name.find('.png')]
    T = 15
    categories = _categories()
    if wild_cate not in categories:
        wild_cate = categories[2]
    small = Config.small_img != 0
    LSTM_hybrid = Config.LSTM_hybrid != 0
    img_dim = SIZE[small]
    output_folder = Config.results_dir
    print('output_folder:', output_folder)
    os.makedirs(output_folder, exist_ok=True)
    vocab_dict = _load_vocab()

    models.reset_default_graph()
    store, _ = models.get_store(Config.block_type, Config.vocab_size, img_dim[0])
    path = latest_checkpoint(Config.ckpt_dir)
    print('Restore trained model:', path)
    restore_checkpoint(store, path)

    class_id = np.array([categories.index(wild_cate)])
    vocab_indices = np.expand_dims(np.array(preprocess_sentence(instruction, vocab_dict, T), dtype=np.int32), axis=0)
    try:
        # uint8 in, uint8 out: normalisation, layout and the truncating cast run on the device (the arithmetic of
        # _normalise / _postprocess, bit for bit: tests/test_gpu_edge_cases.py), the network reads / writes NHWC
        from .. import hip
        tower = models.get_trainer(Config.block_type, Config.vocab_size, img_dim[0])
        tower.G.lstm_hybrid = bool(LSTM_hybrid)
        # the file's pixels go to the device as they are; resize (PIL LANCZOS arithmetic), padding and channel
        # replication of resize_and_padding_mask_image run there (ssc_resample_u8, bit-equal: tests/test_gpu_edge_cases.py)
        from PIL import Image
        from .input_pipeline import resize_and_padding_mask_image_device
        raw = np.array(Image.open(os.path.join(wild_data_base_dir, img_name)).convert('RGB'), dtype=np.uint8)
        sk_dev = torch.from_numpy(np.ascontiguousarray(raw)).cuda()
        if raw.shape[1] != img_dim[0] or raw.shape[0] != img_dim[1]:
            sk_dev = resize_and_padding_mask_image_device(sk_dev, img_dim[0], margin_size=0 if wild_cate in ['road'] else 10)
        sk_u8 = sk_dev[None].contiguous()
        gen_u8 = tower.generate_u8(sk_u8, vocab_indices, torch.randn(1, 256, device='cuda'),
                                   labels=torch.as_tensor(class_id, dtype=torch.int32, device='cuda')).cpu().numpy()
        in_u8 = hip.image_postprocess_u8(hip.sketch_preprocess_u8(sk_u8)).cpu().numpy()
    except Exception as e:      # the reference swallows sess.run errors and prints them (:590-599)
        print(e.args)
        raise
    img_out_filename = img_name[:-4] + '_output.png'
    _write_png(os.path.join(output_folder, img_out_filename), gen_u8[0])
    _write_png(os.path.join(output_folder, img_name[:-4] + '_input.png'), in_u8[0])
    print('Saved file %s' % img_out_filename)


def test():
    """Loop of the inference body over data/captions/<cat>/test.json (main_procedure.py:361-492)."""
    T = 15
    small = Config.small_img != 0
    img_dim = SIZE[small]
    categories = _categories()
    vocab_dict = _load_vocab()
    os.makedirs(Config.results_dir, exist_ok=True)
    models.reset_default_graph()
    store, _ = models.get_store(Config.block_type, Config.vocab_size, img_dim[0])
    restore_checkpoint(store, latest_checkpoint(Config.ckpt_dir))
    for cate in categories:
        cap_path = os.path.join('data', 'captions', cate, 'test.json')
        if not os.path.exists(cap_path):
            continue
        with open(cap_path) as f:
            caps = json.load(f)
        for key, text in (caps.items() if isinstance(caps, dict) else []):
            if isinstance(text, (list, tuple)):
                text = text[0]
            from PIL import Image
            sk = Image.open(os.path.join('data', 'images', cate, 'sketch', key)).convert('RGB')
            margin = 0 if cate in ['road'] else 10
            sketch = resize_and_padding_mask_image(sk, img_dim[0], margin_size=margin)
            if cate in ['house', 'road']:
                sketch = thicken_drawings(sketch)
            x = _normalise(sketch.astype(np.float32))
            idx = np.expand_dims(np.array(preprocess_sentence(text, vocab_dict, T), dtype=np.int32), 0)
            gen, _, ins = build_single_graph(x, x, None, np.array([categories.index(cate)]), None, idx, batch_size=1,
                                             training=False, LSTM_hybrid=Config.LSTM_hybrid != 0,
                                             vocab_size=Config.vocab_size, data_format=Config.data_format,
                                             distance_map=False, block_type=Config.block_type)
            stem = '%s_%s' % (cate, key[:-4])
            _write_png(os.path.join(Config.results_dir, stem + '_output.png'), _postprocess(gen)[0])
            _write_png(os.path.join(Config.results_dir, stem + '_input.png'), _postprocess(ins)[0])


def validation(**kwargs):
    """The reference validates from the data/tfrecord/val queue (main_procedure.py:245-358): here the
    restored generator runs over data/tfrecord/val when that directory exists, otherwise over one seeded synthetic
    batch, and writes validation_results/with_text/<category>_<name>_{output,target,input}.png."""
    small = Config.small_img != 0
    img = SIZE[small][0]
    from ..synthetic import synthetic_batch
    models.reset_default_graph()
    store, _ = models.get_store(Config.block_type, Config.vocab_size, img)
    restore_checkpoint(store, latest_checkpoint(Config.ckpt_dir))
    out_dir = os.path.join(Config.results_dir, 'with_text' if Config.LSTM_hybrid != 0 else 'without_text')
    os.makedirs(out_dir, exist_ok=True)

    def run(images_, sketches_, class_id_, text_, stems):
        gen, images, sketches = build_single_graph(images_, sketches_, None, class_id_, None, text_,
                                                   batch_size=Config.batch_size, training=False,
                                                   LSTM_hybrid=Config.LSTM_hybrid != 0, vocab_size=Config.vocab_size,
                                                   data_format=Config.data_format, distance_map=False,
                                                   block_type=Config.block_type)
        for i, stem in enumerate(stems):
            _write_png(os.path.join(out_dir, stem + '_output.png'), _postprocess(gen)[i])
            _write_png(os.path.join(out_dir, stem + '_target.png'), _postprocess(images)[i])
            _write_png(os.path.join(out_dir, stem + '_input.png'), _postprocess(sketches)[i])

    if os.path.isdir(os.path.join('data', 'tfrecord', 'val')):      # the reference's validation set (:262-272)
        from .input_pipeline import build_input_queue_paired_test
        q = build_input_queue_paired_test('val', Config.batch_size, data_format=Config.data_format, small=small)
        while True:
            try:
                images, sketches, cls, text, cats, names = q.dequeue(with_names=True)
            except StopIteration:
                break
            dev = lambda a: a if torch.is_tensor(a) else torch.from_numpy(a).cuda()
            run(dev(images), dev(sketches), dev(cls), text,
                ['%s_%s' % (c, n[:-4] if n.endswith('.png') else n) for c, n in zip(cats, names)])
        return
    b = synthetic_batch(Config.batch_size, 4321, img, Config.vocab_size)
    cls = b['class_id'].cpu().numpy()
    run(b['images'], b['sketches'], b['class_id'], b['text'],
        ['%s_%04d' % (CATEGORIES[int(cls[i])], i) for i in range(Config.batch_size)])
