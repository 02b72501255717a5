"""Fixtures for the image resampling either side of the generator: inputs and the outputs of the Pillow installed in the
build container (PIL.Image.resize -- what input_pipeline.resize_and_padding_mask_image (ANTIALIAS = LANCZOS) and
scipy.misc.imresize inside Pipeline_utils/fg_color_utils.reverse_resize_image (bilinear) call).  The reference functions
themselves cannot be imported here (tensorflow / cv2 / scipy.misc.imresize are absent), so the fixtures hold PIL's output
for the same calls; run from the repo root:  python tests/golden/make_resize_goldens.py"""
import os

import numpy as np
import PIL
from PIL import Image

rng = np.random.RandomState(20260928)
out = {'pillow_version': np.array(PIL.__version__)}


def sketch(h, w):
    a = np.full((h, w), 255, np.uint8)
    for _ in range(12):
        y, x = rng.randint(0, h), rng.randint(0, w)
        a[max(0, y - 1):y + 2, max(0, x - 30):x + 30] = rng.randint(0, 120)
        a[max(0, y - 25):y + 25, max(0, x - 1):x + 2] = rng.randint(0, 120)
    return np.repeat(a[:, :, None], 3, axis=2)


cases = [(sketch(300, 260), 'lanczos', (173, 150)), (rng.randint(0, 256, (64, 80, 3)).astype(np.uint8), 'lanczos', (192, 240)),
         (rng.randint(0, 256, (250, 111, 3)).astype(np.uint8), 'lanczos', (97, 43)),
         (rng.randint(0, 256, (192, 150, 3)).astype(np.uint8), 'bilinear', (61, 220)),
         (rng.randint(0, 256, (120, 192, 3)).astype(np.uint8), 'bilinear', (330, 192))]
for i, (src, filt, (nh, nw)) in enumerate(cases):
    res = Image.fromarray(src).resize((nw, nh), resample={'lanczos': Image.LANCZOS, 'bilinear': Image.BILINEAR}[filt])
    out['src_%d' % i], out['filt_%d' % i], out['out_%d' % i] = src, np.array(filt), np.array(res)
out['n_cases'] = np.array(len(cases))


def resize_and_padding_mask_image(image, new_size, margin_size):     # input_pipeline.py:199-239 with PIL doing the resize
    height, width = image.height + margin_size * 2, image.width + margin_size * 2
    scale = new_size / max(height, width)
    new_h, new_w = int(round(image.height * scale)), int(round(image.width * scale))
    if scale != 1:
        image = image.resize((new_w, new_h), resample=Image.LANCZOS)
    img_np = np.array(image, dtype=np.uint8)[:, :, 0]
    top, left = (new_size - new_h) // 2, (new_size - new_w) // 2
    rst = np.pad(img_np, [(top, new_size - new_h - top), (left, new_size - new_w - left)], mode='constant', constant_values=255)
    return np.repeat(rst[:, :, None], 3, axis=2)


pads = [(sketch(300, 260), 192, 10), (sketch(150, 420), 192, 0), (sketch(90, 70), 64, 10), (sketch(172, 172), 192, 10)]
for i, (src, size, margin) in enumerate(pads):
    out['pad_src_%d' % i], out['pad_size_%d' % i], out['pad_margin_%d' % i] = src, np.array(size), np.array(margin)
    out['pad_out_%d' % i] = resize_and_padding_mask_image(Image.fromarray(src), size, margin)
out['n_pad'] = np.array(len(pads))


def reverse_resize_image(inst, box_h, box_w, margin_size):      # fg_color_utils.py:137-160, imresize = PIL bilinear
    s = inst.shape[0]
    bh, bw = box_h + 2 * margin_size, box_w + 2 * margin_size
    if bh > bw:
        pad = int(round(s * (bh - bw) / bh / 2.))
        cut = inst[:, pad:s - pad]
    else:
        pad = int(round(s * (bw - bh) / bw / 2.))
        cut = inst[pad:s - pad, :]
    rev = np.array(Image.fromarray(np.ascontiguousarray(cut)).resize((bw, bh), resample=Image.BILINEAR))
    return rev[margin_size:margin_size + box_h, margin_size:margin_size + box_w]


revs = [(rng.randint(0, 256, (192, 192, 3)).astype(np.uint8), 300, 180, 10), (rng.randint(0, 256, (192, 192, 3)).astype(np.uint8), 90, 260, 0),
        (rng.randint(0, 256, (64, 64, 3)).astype(np.uint8), 40, 41, 10)]
for i, (src, bh, bw, margin) in enumerate(revs):
    out['rev_src_%d' % i], out['rev_bh_%d' % i], out['rev_bw_%d' % i], out['rev_margin_%d' % i] = src, np.array(bh), np.array(bw), np.array(margin)
    out['rev_out_%d' % i] = reverse_resize_image(src, bh, bw, margin)
out['n_rev'] = np.array(len(revs))
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'resize_goldens.npz'), **out)
print('wrote resize_goldens.npz:', len(cases), 'resize,', len(pads), 'pad,', len(revs), 'reverse cases; Pillow', PIL.__version__)
