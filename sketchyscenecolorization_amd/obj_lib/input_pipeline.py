"""Host side of obj_lib/input_pipeline.py: num_classes (:11-15), the paired TFRecord input queue (:43-154, read
without TensorFlow through sketchyscenecolorization_amd.tfrecord), split_inputs (:184-196) and the sketch
pre-processing of the inference path (:199-257)."""
import os
import random

import numpy as np

num_classes = 25


def get_num_classes():
    return num_classes


def split_inputs(input_data, batch_size, batch_portion, num_gpu):
    """Contiguous per-tower slices of the global batch (tower i gets batch_size*batch_portion[i] samples)."""
    out, start = [], 0
    for i in range(num_gpu):
        size = int(batch_size * batch_portion[i])
        out.append(input_data[start:start + size])
        start += size
    return out


def resize_and_padding_mask_image(image, new_size, resample_method=None, margin_size=10):
    """PIL image -> [new_size,new_size,3] uint8: scale the longer side (plus margins) to new_size with
    ANTIALIAS(=LANCZOS), centre it on a white canvas, replicate channel 0 (input_pipeline.py:199-239)."""
    from PIL import Image
    if resample_method is None:
        resample_method = Image.LANCZOS
    height = image.height + margin_size * 2
    width = image.width + margin_size * 2
    scale = new_size / max(height, width)
    new_h = int(round(image.height * scale))
    new_w = int(round(image.width * scale))
    assert new_h <= new_size and new_w <= new_size
    if scale != 1:
        image = image.resize((new_w, new_h), resample=resample_method)
    img_np = np.array(image, dtype=np.uint8)[:, :, 0]
    top = (new_size - new_h) // 2
    left = (new_size - new_w) // 2
    canvas = np.pad(img_np, [(top, new_size - new_h - top), (left, new_size - new_w - left)], mode='constant',
                    constant_values=255)
    assert canvas.shape == (new_size, new_size)
    return np.repeat(canvas[:, :, None], 3, axis=2)


# ---------------------------------------------------------------------------------------------------------------
# the same on the device: Pillow's 8-bit resampler as HIP kernels (ssc_resample_u8), coefficient tables from the host
# ---------------------------------------------------------------------------------------------------------------
_PRECISION_BITS = 32 - 8 - 2


def resample_coeffs(in_size, out_size, filt='lanczos'):
    """Bounds and 22-bit fixed-point coefficients of one axis of PIL.Image.resize for 8-bit images (Pillow
    libImaging/Resample.c: precompute_coeffs + normalize_coeffs_8bpc), box = the whole axis.
    -> (bounds int32 [out,2] = {first tap, taps}, coefficients int32 [out,ksize])."""
    support = {'lanczos': 3.0, 'bilinear': 1.0}[filt]
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support *= filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    center = (np.arange(out_size, dtype=np.float64) + 0.5) * scale
    xmin = np.maximum((center - support + 0.5).astype(np.int64), 0)           # C int(): the values are >= -support
    xmax = np.minimum((center + support + 0.5).astype(np.int64), in_size) - xmin
    x = np.arange(ksize, dtype=np.float64)[None, :]
    arg = (x + xmin[:, None] - center[:, None] + 0.5) / filterscale
    if filt == 'lanczos':
        with np.errstate(divide='ignore', invalid='ignore'):
            a = np.pi * arg
            w = np.where(arg == 0.0, 1.0, (np.sin(a) / a) * (np.sin(a / 3.0) / (a / 3.0)))
        w = np.where((arg >= -3.0) & (arg < 3.0), w, 0.0)
    else:
        w = np.where(np.abs(arg) < 1.0, 1.0 - np.abs(arg), 0.0)
    w = np.where(x < xmax[:, None], w, 0.0)
    ww = w.sum(axis=1, keepdims=True)
    w = np.where(ww != 0.0, w / np.where(ww != 0.0, ww, 1.0), w)
    kk = np.trunc(w * (1 << _PRECISION_BITS) + np.where(w >= 0, 0.5, -0.5)).astype(np.int32)
    return np.stack([xmin, xmax], axis=1).astype(np.int32), kk


def _dev_coeffs(in_size, out_size, filt):
    import torch
    if in_size == out_size:
        return None
    b, k = resample_coeffs(in_size, out_size, filt)
    return torch.from_numpy(b).cuda(), torch.from_numpy(np.ascontiguousarray(k)).cuda()


def resize_and_padding_mask_image_device(img_u8, new_size, margin_size=10):
    """resize_and_padding_mask_image on the device: img_u8 uint8 [H,W,3] (device tensor) -> uint8 [new_size,new_size,3]
    (device), the PIL LANCZOS resize of channel 0, centred on a white canvas, replicated over 3 channels -- bit for bit what
    the host function above returns."""
    from .. import hip
    h, w = int(img_u8.shape[0]), int(img_u8.shape[1])
    scale = new_size / max(h + 2 * margin_size, w + 2 * margin_size)
    new_h, new_w = int(round(h * scale)), int(round(w * scale))
    assert new_h <= new_size and new_w <= new_size
    if scale == 1:
        new_h, new_w = h, w
    top, left = (new_size - new_h) // 2, (new_size - new_w) // 2
    return hip.resample_u8(img_u8, new_h, new_w, _dev_coeffs(w, new_w, 'lanczos'), _dev_coeffs(h, new_h, 'lanczos'), chan=0,
                           out_hw=(new_size, new_size), top=top, left=left, fill=255, out_channels=3)


def reverse_resize_image_device(inst_u8, box_h, box_w, h_w_ratio=1, margin_size=10):
    """Pipeline_utils/fg_color_utils.py:137-160 on the device: cut the padding of the generated [S,S,3] instance, bilinear
    resize (scipy.misc.imresize = PIL) to the box plus margins, cut the margins -> uint8 [box_h, box_w, 3] (device)."""
    from .. import hip
    s = int(inst_u8.shape[0])
    bh, bw = box_h + 2 * margin_size, box_w + 2 * margin_size
    if bh * h_w_ratio > bw:
        pad = int(round(s * (bh * h_w_ratio - bw) / (bh * h_w_ratio) / 2.))
        cut = inst_u8[:, pad:s - pad]
    else:
        pad = int(round(s * (bw - bh * h_w_ratio) / bw / 2.))
        cut = inst_u8[pad:s - pad, :]
    cut = cut.contiguous()
    ch, cw = int(cut.shape[0]), int(cut.shape[1])
    rev = hip.resample_u8(cut, bh, bw, _dev_coeffs(cw, bw, 'bilinear'), _dev_coeffs(ch, bh, 'bilinear'))
    return rev[margin_size:margin_size + box_h, margin_size:margin_size + box_w].contiguous()


def thicken_drawings(image):
    """2x2 grey dilation of the (dark) strokes (input_pipeline.py:242-257 calls
    skimage.morphology.dilation(img, square(2))).  skimage pads an even footprint with a zero row/column at
    the start, so the neighbourhood of pixel (i,j) is rows {i,i+1} x cols {j,j+1}; skimage is not installed
    here, so this offset convention is restated from its source, not verified by execution."""
    img = 255 - np.array(image[:, :, 0], dtype=np.uint8)
    p = np.pad(img, ((0, 1), (0, 1)), mode='edge')
    dil = np.maximum(np.maximum(p[:-1, :-1], p[1:, :-1]), np.maximum(p[:-1, 1:], p[1:, 1:]))
    dil = 255 - dil
    return np.repeat(dil[:, :, None], 3, axis=2).astype(np.uint8)


# ---------------------------------------------------------------------------------------------------------------
# paired TFRecord queue (get_paired_input + build_input_queue_paired, :43-154)
# ---------------------------------------------------------------------------------------------------------------
SIZE = {True: (64, 64), False: (192, 192)}
RECORD_HW = 384         # "cannot change": the records hold 384x384x3 uint8 images (:77, 82)
T_STEPS = 15


def decode_paired_example(feat, img_dim, rng, data_format='NCHW', distance_map=False):
    """One parsed Example -> (image, sketch) float32 in [-1,1] (NCHW), class id, caption indices [15].

    tf.image.resize_images with the TF1 defaults (align_corners=False, no half-pixel centres) maps output pixel i to
    source coordinate i * (384 / size): for the integer factors 2 (192) and 6 (64) BILINEAR is exactly the source pixel
    at that coordinate and AREA is the mean of the factor x factor block."""
    img = np.frombuffer(feat['cartoon_data'][0], dtype=np.uint8).astype(np.float32).reshape(RECORD_HW, RECORD_HW, 3)
    sk = np.frombuffer(feat['sketch_data'][0], dtype=np.uint8).astype(np.float32).reshape(RECORD_HW, RECORD_HW, 3)
    if distance_map:        # :86-96: binarise at 250, Euclidean distance to the nearest stroke pixel, scaled to [0, 255]
        from scipy import ndimage
        sk = np.where(sk < 250, 0.0, 255.0).astype(np.float32)
        sk = ndimage.distance_transform_edt(sk).astype(np.float32)
        sk = sk / sk.max() * 255.0
    size = img_dim[0]
    if size != RECORD_HW:
        f = RECORD_HW // size
        assert f * size == RECORD_HW
        img = img[::f, ::f]
        sk = sk.reshape(size, f, size, f, 3).mean(axis=(1, 3))
    img = (img - img.min()) / (img.max() - img.min() + 1)
    img = img + rng.uniform(0.0, 1.0 / 256, size=img.shape).astype(np.float32)      # dequantisation noise (:117)
    img = img * 2.0 - 1.0
    sk = sk / 255.0 * 2.0 - 1.0
    if data_format == 'NCHW':
        img, sk = img.transpose(2, 0, 1), sk.transpose(2, 0, 1)
    text = np.frombuffer(feat['Text_vocab_indices'][0], dtype=np.uint8).astype(np.int32).reshape(T_STEPS)
    return (np.ascontiguousarray(img, dtype=np.float32), np.ascontiguousarray(sk, dtype=np.float32),
            int(feat['Category_id'][0]), text, feat.get('Category', [b''])[0].decode('utf-8', 'replace'),
            feat.get('ImageName', [b''])[0].decode('utf-8', 'replace'))


class PairedQueue(object):
    """build_input_queue_paired: examples of data/tfrecord/<mode>/* in shuffled order, endlessly (num_epochs=None),
    through a shuffle buffer of ``min_after_dequeue`` decoded examples (tf.train.maybe_shuffle_batch, :143-148)."""

    def __init__(self, mode, batch_size, data_format='NCHW', distance_map=False, small=False, min_after_dequeue=512,
                 data_base_dir='data', seed=None, device_decode=None, prefetch=None):
        """device_decode (default: on when a GPU is there and the layout is NCHW): the shuffle buffer keeps the raw
        uint8 records and a batch is resized / normalised by one kernel at dequeue (hip.decode_paired_u8, the arithmetic
        of decode_paired_example bit for bit); ``dequeue`` then returns device tensors.  The host decode costs ~3 ms per
        example, ten times the GPU's step time at batch 32."""
        from .. import tfrecord
        assert mode in ('train', 'val', 'test')
        data_dir = os.path.join(data_base_dir, 'tfrecord', mode)
        self.files = sorted(os.path.join(data_dir, f) for f in os.listdir(data_dir)
                            if os.path.isfile(os.path.join(data_dir, f)))
        print('build_input_queue_paired from %s: paired file num: %d' % (data_dir, len(self.files)))
        self.tf, self.batch_size, self.fmt, self.dm = tfrecord, batch_size, data_format, distance_map
        self.img_dim = SIZE[bool(small)]
        self.shuffle = mode == 'train'
        self.min_after = min_after_dequeue if self.shuffle else 0
        self.rng = random.Random(seed)
        self.np_rng = np.random.RandomState(self.rng.randrange(2 ** 31))
        self.buf = []
        if device_decode is None:
            import torch
            device_decode = torch.cuda.is_available() and data_format == 'NCHW'
        self.device_decode = bool(device_decode)
        self._gen, self._gen_seed = None, self.rng.randrange(2 ** 31)   # drawn in both modes: same example order
        self._it = self._examples()
        # Training from records: the host half of a batch (read, CRC, parse, stage: ~0.4 ms per 884 KB record, 64 records per
        # iteration of a 12 ms step) runs one batch ahead on a thread of its own.  SSC_RECORD_PREFETCH=0: in the caller.
        # (asked for by the training procedure, main_procedure.RecordQueue; a queue built directly stays synchronous)
        self.prefetch = bool(prefetch) and self.device_decode and self.shuffle
        self._ring, self._ring_i, self._q, self._thread, self._stop = None, 0, None, None, False

    def _examples(self):
        while True:
            files = list(self.files)
            if self.shuffle:
                self.rng.shuffle(files)
            for path in files:
                if self.device_decode:
                    # records as views of the mapped file: CRC in place, the images copied once (into the staging buffer)
                    for rec in self.tf.read_records(path, views=True):
                        yield self._raw_example(self.tf.parse_example(rec, views=True))
                    continue
                for rec in self.tf.read_records(path):
                    feat = self.tf.parse_example(rec)
                    yield decode_paired_example(feat, self.img_dim, self.np_rng, self.fmt, self.dm)
            if not self.shuffle:
                return

    @staticmethod
    def _raw_example(feat):
        """The undecoded fields of one Example: (image bytes, sketch bytes, class id, caption, category, name)."""
        text = np.frombuffer(feat['Text_vocab_indices'][0], dtype=np.uint8).astype(np.int32).reshape(T_STEPS)
        return (feat['cartoon_data'][0], feat['sketch_data'][0], int(feat['Category_id'][0]), text,
                feat.get('Category', [b''])[0].decode('utf-8', 'replace'),
                feat.get('ImageName', [b''])[0].decode('utf-8', 'replace'))

    # ---- device decode: host half (any thread) and device half (the caller's thread and stream) ----
    _RING = 3       # pinned staging buffers per queue: one being filled, one in flight to the device, one spare

    def _stage(self, ex):
        """Host half: the raw images of a batch into a pinned staging buffer [2, n, 384, 384, 3] (uint8).  The buffers are
        allocated once -- a fresh 28 MB array per batch is page-faulted in every time -- and reused in turn once the copy that
        read them has happened."""
        import torch
        n = len(ex)
        if self._ring is None or self._ring[0][0].shape[1] != n:
            self._ring = [[torch.empty((2, n, RECORD_HW, RECORD_HW, 3), dtype=torch.uint8).pin_memory(), None]
                          for _ in range(self._RING)]
            self._ring_i = 0
        slot = self._ring[self._ring_i % self._RING]
        self._ring_i += 1
        if slot[1] is not None:
            slot[1].synchronize()       # the device copy out of this buffer (three batches ago) has happened
            slot[1] = None
        raw = slot[0].numpy()
        for k, e in enumerate(ex):
            raw[0, k] = np.frombuffer(e[0], dtype=np.uint8).reshape(RECORD_HW, RECORD_HW, 3)
            raw[1, k] = np.frombuffer(e[1], dtype=np.uint8).reshape(RECORD_HW, RECORD_HW, 3)
        return slot

    def _decode_on_device(self, ex, slot=None):
        import torch
        from .. import hip
        n, size = len(ex), self.img_dim[0]
        slot = slot if slot is not None else self._stage(ex)
        dev = slot[0].to('cuda', non_blocking=True)
        slot[1] = torch.cuda.Event()
        slot[1].record()
        if self._gen is None:
            self._gen = torch.Generator(device='cuda')
            self._gen.manual_seed(self._gen_seed)
        noise = torch.rand((n, size, size, 3), device='cuda', generator=self._gen) * (1.0 / 256)   # dequantisation (:117)
        return hip.decode_paired_u8(dev[0], dev[1], size, noise=noise, distance_map=self.dm)

    # ---- prefetch: the host half of the NEXT batches on a thread of its own ----
    def _producer(self):
        """Examples are drawn, and their bytes staged, strictly in dequeue order by this one thread -- the queue's order and
        its random numbers are those of the synchronous queue.  File reads, the CRC (a ctypes call) and the large copies
        release the interpreter lock, so the training loop's own host work goes on beside them."""
        import queue
        import torch

        def put(item):
            while not self._stop:
                try:
                    self._q.put(item, timeout=0.2)
                    return True
                except queue.Full:
                    continue
            return False

        try:
            torch.cuda.set_device(self._device)     # (the current device is a per-thread setting)
            while not self._stop:
                ex = []
                for _ in range(self.batch_size):
                    try:
                        ex.append(self._next())
                    except StopIteration:
                        break
                if len(ex) < self.batch_size:
                    put(StopIteration())
                    return
                if not put((ex, self._stage(ex))):
                    return
        except BaseException as e:       # whatever went wrong is raised in the consumer
            put(e)

    def _start_prefetch(self):
        import queue
        import threading
        import torch
        self._device = torch.cuda.current_device()
        self._q = queue.Queue(maxsize=self._RING - 2)       # staged batches waiting: never more than the ring can hold apart
        self._thread = threading.Thread(target=self._producer, name='PairedQueue-prefetch', daemon=True)
        self._thread.start()

    def close(self):
        """Stop the prefetch thread (it otherwise lives, blocked on its queue, until the process ends)."""
        self._stop = True

    def _next(self):
        while len(self.buf) <= self.min_after:
            try:
                self.buf.append(next(self._it))
            except StopIteration:
                break
        if not self.buf:
            raise StopIteration
        i = self.rng.randrange(len(self.buf)) if self.shuffle else 0
        return self.buf.pop(i)

    def dequeue(self, with_names=False):
        """One batch: (images [N,3,h,w], sketches, class ids int32 [N], caption indices int32 [N,15])
        [+ category names, image names].  Raises StopIteration when a val / test epoch is exhausted."""
        if self.prefetch:
            if self._thread is None:
                self._start_prefetch()
            item = self._q.get()
            if isinstance(item, BaseException):
                self._q.put(item)           # (a later call meets it again)
                raise StopIteration if isinstance(item, StopIteration) else item
            ex, slot = item
            images, sketches = self._decode_on_device(ex, slot)
            out = (images, sketches, np.array([e[2] for e in ex], dtype=np.int32), np.stack([e[3] for e in ex]))
            return out + ([e[4] for e in ex], [e[5] for e in ex]) if with_names else out
        ex = []
        for _ in range(self.batch_size):
            try:
                ex.append(self._next())
            except StopIteration:
                break
        if len(ex) < self.batch_size:       # tf.train.maybe_batch drops the incomplete final batch
            raise StopIteration
        if self.device_decode:
            images, sketches = self._decode_on_device(ex)
        else:
            images, sketches = np.stack([e[0] for e in ex]), np.stack([e[1] for e in ex])
        out = (images, sketches, np.array([e[2] for e in ex], dtype=np.int32), np.stack([e[3] for e in ex]))
        return out + ([e[4] for e in ex], [e[5] for e in ex]) if with_names else out


def build_input_queue_paired_test(mode, batch_size, data_format='NCHW', distance_map=False, small=False, one_hot=False,
                                  capacity=8192, data_base_dir='data'):
    """Reference signature (:157-181): one unshuffled epoch of data/tfrecord/<val|test>."""
    assert mode in ['test', 'val'] and not one_hot
    return PairedQueue(mode, batch_size, data_format, distance_map, small, 0, data_base_dir)


def build_input_queue_paired(mode, batch_size, data_format='NCHW', distance_map=False, small=False, one_hot=False,
                             capacity=8192, min_after_dequeue=512, data_base_dir='data'):
    """Reference signature (:131-154).  Returns the queue; ``queue.dequeue()`` yields what the reference's seven
    tensors carry that the model uses (images, sketches, dense labels, caption indices)."""
    assert mode in ['train'] and not one_hot
    return PairedQueue(mode, batch_size, data_format, distance_map, small, min_after_dequeue, data_base_dir)
