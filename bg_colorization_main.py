#!/usr/bin/env python
"""Command line of the Background_Colorization module on the MI355X-native HIP path.

Same flags and defaults as the reference (bg_colorization_main.py:978-1003), same run directory
``outputs/<UTC stamp>/{snapshot,log,results}``, snapshots named ``snapshot-<global step>``, test mode writing
``<bg name>_{inputs,outputs,targets}.png`` with the foreground pasted back over the generated background (:861-871).

Data layout under --data_base_dir (reference :739-750): ``foreground/<mode>/*.png``, ``background/<mode>/*.png``,
``segment/<mode>/*.png`` and ``captions/<mode>.json`` (records with fg_name, bg_name, color_text).  When the caption
file is missing the run uses seeded synthetic scenes, so the CLI can be exercised without the dataset.
"""
import argparse
import json
import os

os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')     # kernel arguments in device memory (measured: 1570 vs 1540 images/s with 0)
import random
import time

import numpy as np
import torch

FLAGS = [
    ('mode', str, 'train', ['train', 'test'], 'train or test'),
    ('resume_from', str, '', None, 'stamp of an earlier run under outputs/'),
    ('data_base_dir', str, 'data', None, 'dataset root'),
    ('image_size', int, 768, None, 'square image size'),
    ('batch_size', int, 1, None, 'images per step (the reference graph is built for 1)'),
    ('max_steps', int, 100000, None, 'training steps'),
    ('lr', float, 0.0002, None, 'initial Adam step size'),
    ('l1_weight', float, 100.0, None, 'weight of the masked L1 term'),
    ('gan_weight', float, 1.0, None, 'weight of the GAN term'),
    ('seg_weight', float, 100.0, None, 'weight of the region-mask term'),
    ('seg_classes', int, 3, None, 'region classes'),
    ('ngf', int, 64, None, 'generator width'),
    ('ndf', int, 64, None, 'discriminator width'),
    ('text_len', int, 8, None, 'caption length'),
    ('vocab_size', int, 18, None, 'caption vocabulary size'),
    ('vocab_file', str, 'data/bg_vocab.txt', None, 'vocabulary file'),
    ('summary_freq', int, 200, None, 'steps between scalar summaries'),
    ('progress_freq', int, 50, None, 'steps between progress prints'),
    ('save_freq', int, 20000, None, 'steps between snapshots (0 = never)'),
]


def build_parser():
    parser = argparse.ArgumentParser(description=__doc__.split('\n')[0])
    for name, typ, default, choices, text in FLAGS:
        parser.add_argument('--' + name, type=typ, default=default, choices=choices, help=text)
    return parser


class Scenes(object):
    """The four arrays of one training example: (inputs u8 [1,H,W,3], targets u8, token ids [1,T], labels [1,H,W])."""

    def __init__(self, p):
        self.p = p
        self.size, self.T = p['image_size'], p['text_len']
        base, mode = p['data_base_dir'], p['mode']
        self.dirs = {k: os.path.join(base, k, mode) for k in ('foreground', 'background', 'segment')}
        cap = os.path.join(base, 'captions', mode + '.json')
        self.records, self.vocab = None, None
        if os.path.exists(cap):
            from sketchyscenecolorization_amd.data_processing.text_processing import load_vocab_dict_from_file
            with open(cap) as fp:
                self.records = json.load(fp)
            self.vocab = load_vocab_dict_from_file(p['vocab_file'])
        print('## nImgs =', len(self), '\n')

    def __len__(self):
        return len(self.records) if self.records is not None else 8

    def get(self, idx, is_test=False):
        size = self.size
        if self.records is None:        # seeded synthetic scene
            rng = np.random.RandomState(1000 + idx)
            fg = np.full((1, size, size, 3), 255, np.uint8)
            fg[0, size // 4:size // 2, size // 4:size // 2] = rng.randint(0, 255, 3)
            bg = rng.randint(0, 255, (1, size, size, 3)).astype(np.uint8)
            lab = np.zeros((1, size, size), np.int32)
            lab[0, :size // 2] = 1
            lab[0, size // 2:] = 2
            lab[0, size // 4:size // 2, size // 4:size // 2] = 0
            tok = np.zeros((1, self.T), np.int32)
            tok[0, self.T - 4:] = rng.randint(1, self.p['vocab_size'], 4)
            return fg, bg, tok, (np.zeros_like(lab) if is_test else lab), 'synthetic_%d.png' % idx, 'synthetic_%d.png' % idx
        from sketchyscenecolorization_amd.data_processing.image_processing import load_image, load_region_mask
        from sketchyscenecolorization_amd.data_processing.text_processing import preprocess_sentence
        rec = self.records[idx]
        fg = load_image(os.path.join(self.dirs['foreground'], rec['fg_name']), size)
        bg = load_image(os.path.join(self.dirs['background'], rec['bg_name']), size)
        tok = np.array(preprocess_sentence(rec['color_text'], self.vocab, self.T), dtype=np.int32)[None]
        # segment png: 0 = foreground, 128 = sky (1), 255 = ground (2)   (image_processing.py:14-24)
        lab = load_region_mask(os.path.join(self.dirs['segment'], rec['fg_name']), size, is_test)
        return fg, bg, tok, lab, rec['fg_name'], rec['bg_name']


def to_unit(u8):
    """uint8 [0,255] -> float [-1,1] (convert_image_dtype + preprocess, :30-33, 100-113)."""
    u8 = u8 if torch.is_tensor(u8) else torch.from_numpy(u8)
    return u8.to('cuda', torch.float32) / 255.0 * 2.0 - 1.0


def to_u8(x):
    """[-1,1] -> uint8 with saturation (deprocess + convert_image_dtype(saturate=True), :36-39, 785-786)."""
    y = ((x + 1.0) / 2.0).clamp(0.0, 1.0) * 255.0
    return (y + 0.5).floor().clamp(0, 255).to(torch.uint8).cpu().numpy()


def bg_colorization(**p):
    from sketchyscenecolorization_amd.bg_colorization import BGTrainer
    mode, stamp = p['mode'], p['resume_from']
    if stamp == '':
        if mode == 'test':
            raise Exception('checkpoint required for test mode')
        stamp = time.strftime('%Y-%m-%d-%H-%M-%S', time.gmtime())
    out_dir = os.path.join('outputs', stamp)
    snap_dir = os.path.join(out_dir, 'snapshot')
    os.makedirs(snap_dir, exist_ok=True)
    scenes = Scenes(p)
    tr = BGTrainer(image_size=p['image_size'], vocab_size=p['vocab_size'], ngf=p['ngf'], ndf=p['ndf'],
                   seg_classes=p['seg_classes'], lr=p['lr'], max_steps=p['max_steps'], gan_weight=p['gan_weight'],
                   l1_weight=p['l1_weight'], seg_weight=p['seg_weight'], seed=random.randint(0, 2 ** 31 - 1))
    print('parameter_count =', tr.store.parameter_count('generator') + tr.store.parameter_count('discriminator'))
    iter_from = 0
    if p['resume_from'] != '':
        idx = os.path.join(snap_dir, 'checkpoint')
        with open(idx) as fp:
            name = fp.readline().split('"')[1]
        print('loading model from checkpoint', os.path.join(snap_dir, name))
        from sketchyscenecolorization_amd import tf_checkpoint
        if tf_checkpoint.is_tf_checkpoint(os.path.join(snap_dir, name)):    # a tf.train.Saver checkpoint (released model)
            tr.store.load_dict(tf_checkpoint.read_checkpoint(os.path.join(snap_dir, name)))
        else:
            sd = torch.load(os.path.join(snap_dir, name), map_location='cpu')
            tr.store.load_state_dict(sd)
            for sc in (tr.store.generator, tr.store.discriminator):
                if '__adam_m__/' + sc.name in sd:
                    sc.adam_m.copy_(sd['__adam_m__/' + sc.name])
        iter_from = int(name.split('-')[1])
        tr.global_step = iter_from
    print('iter_from', iter_from)

    if mode == 'test':
        from PIL import Image
        res_dir = os.path.join(out_dir, 'results')
        os.makedirs(res_dir, exist_ok=True)
        for i in range(len(scenes)):
            print('Processing', i, '/', len(scenes))
            fg, bg, tok, lab, fg_name, bg_name = scenes.get(i, is_test=True)
            gctx = tr.G.forward(to_unit(fg), tok, None, 'bg')
            out = to_u8(gctx['image'])
            seg_path = os.path.join(scenes.dirs['segment'], fg_name)
            if os.path.exists(seg_path):        # paste the foreground (segment value 0) back over the generation
                inner = np.array(Image.open(seg_path).convert('RGB'), np.uint8)[:, :, 0]
                out[0][inner == 0] = fg[0][inner == 0]
            for kind, arr in (('inputs', fg), ('outputs', out), ('targets', bg)):
                Image.fromarray(arr[0], 'RGB').save(os.path.join(res_dir, bg_name[:-4] + '_' + kind + '.png'), 'PNG')
        return

    log_dir = os.path.join(out_dir, 'log')
    os.makedirs(log_dir, exist_ok=True)
    start = time.time()
    ema = None
    # The examples of the next steps are loaded ahead (two 768 x 768 images and a region mask per step: ~60 ms of decoding on
    # one thread against a 22 ms device step) by a few threads, in the order the indices are drawn -- one draw per step, as
    # before; SSC_BG_PREFETCH=0: loaded where they are used.
    import collections
    from concurrent.futures import ThreadPoolExecutor
    depth = int(os.environ.get('SSC_BG_PREFETCH', '4'))
    pool = ThreadPoolExecutor(max_workers=max(depth, 1)) if depth > 0 else None
    ahead, drawn = collections.deque(), [iter_from]

    def draw_more():
        while pool is not None and len(ahead) < depth and drawn[0] < p['max_steps']:
            ahead.append(pool.submit(scenes.get, random.randint(0, len(scenes) - 1)))
            drawn[0] += 1

    # host arrays go to the device through a ring of pinned staging buffers allocated once (pinning a fresh 1.7 MB array costs
    # 3.5 ms a time; a copy from pageable memory returns only when it has happened, behind the step that is running)
    ring, ring_i = {}, [0]

    def pinned(a, slot):
        key = (slot, a.shape, a.dtype.str)
        if key not in ring:
            ring[key] = [torch.from_numpy(np.empty_like(a)).pin_memory() for _ in range(4)]
        buf = ring[key][ring_i[0] % 4]
        buf.numpy()[...] = a
        return buf.to('cuda', non_blocking=True)

    for step in range(iter_from, p['max_steps']):
        def should(freq):
            return freq > 0 and ((step + 1) % freq == 0 or step == p['max_steps'] - 1)
        if pool is not None:
            draw_more()
            fg, bg, tok, lab, _, _ = ahead.popleft().result()
            draw_more()
        else:
            fg, bg, tok, lab, _, _ = scenes.get(random.randint(0, len(scenes) - 1))
        ring_i[0] += 1
        tr.train_step(to_unit(pinned(fg, 'fg')), to_unit(pinned(bg, 'bg')), tok, pinned(lab, 'lab'))
        if should(p['progress_freq']) or should(p['summary_freq']):
            vals = tr.loss_values()
            # tf.train.ExponentialMovingAverage(0.99) of the five losses (:657-658), updated when they are read
            ema = list(vals) if ema is None else [0.99 * e + 0.01 * v for e, v in zip(ema, vals)]
            names = ('discrim_loss', 'gen_loss', 'gen_loss_GAN', 'gen_loss_L1', 'region_mask_loss')
            if should(p['summary_freq']):
                with open(os.path.join(log_dir, 'scalars.jsonl'), 'a') as fp:
                    fp.write(json.dumps(dict(zip(names, ema), step=tr.global_step)) + '\n')
            if should(p['progress_freq']):
                rate = (step - iter_from + 1) * p['batch_size'] / (time.time() - start)
                left = (p['max_steps'] - step) * p['batch_size'] / rate
                print('progress step %d  image/sec %0.1f  left time:%dd %dh %dm'
                      % (tr.global_step, rate, left // 86400, left % 86400 // 3600, left % 3600 // 60))
                for n, v in zip(names, ema):
                    print(n, v)
        if should(p['save_freq']):
            print('saving model to', snap_dir)
            name = 'snapshot-%d' % tr.global_step
            sd = tr.store.state_dict()
            for sc in (tr.store.generator, tr.store.discriminator):
                sd['__adam_m__/' + sc.name] = sc.adam_m.detach().cpu()
            torch.save(sd, os.path.join(snap_dir, name))
            with open(os.path.join(snap_dir, 'checkpoint'), 'w') as fp:
                fp.write('model_checkpoint_path: "%s"\n' % name)


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.mode == 'test':
        assert args.resume_from != ''
    if args.batch_size != 1:
        raise NotImplementedError('the reference graph feeds one image per step (placeholders of batch 1, :765-768)')
    bg_colorization(**{name: getattr(args, name) for name, _t, _d, _c, _h in FLAGS})


if __name__ == '__main__':
    main()
