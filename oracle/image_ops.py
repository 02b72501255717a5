"""CPU restatement (NumPy) of the image pre / post-processing either side of the generator -- TEST INFRASTRUCTURE, never
imported by the product:

  resample_u8                      Pillow's ImagingResample for 8-bit images (libImaging/Resample.c: precompute_coeffs,
                                   normalize_coeffs_8bpc, horizontal then vertical pass with an 8-bit intermediate), which is
                                   what PIL.Image.resize(..., ANTIALIAS = LANCZOS) of input_pipeline.py:199-239 and
                                   scipy.misc.imresize(..., 'bilinear') of Pipeline_utils/fg_color_utils.py:137-160 execute.
                                   The reference does not vendor Pillow; the algorithm is restated from Pillow's published
                                   source and pinned by fixtures generated with the Pillow installed here
                                   (tests/golden/make_resize_goldens.py).
  resize_and_padding_mask_image    input_pipeline.py:199-239
  reverse_resize_image             Pipeline_utils/fg_color_utils.py:137-160
  thicken_drawings                 input_pipeline.py:242-257 (skimage.morphology.dilation(img, square(2)))
  sketch_preprocess / image_postprocess   main_procedure.py:577-610 (x/255*2-1; ((x+1)/2*255).astype(uint8))
  decode_paired_example            input_pipeline.py:77-131 at the integer resize factors of the dataset
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def _lanczos(x):
    if -3.0 <= x < 3.0:
        if x == 0.0:
            return 1.0
        a = math.pi * x
        return (math.sin(a) / a) * (math.sin(a / 3.0) / (a / 3.0))
    return 0.0


def _bilinear(x):
    x = abs(x)
    return 1.0 - x if x < 1.0 else 0.0


FILTERS = {'lanczos': (_lanczos, 3.0), 'bilinear': (_bilinear, 1.0)}


def precompute_coeffs(in_size, out_size, filt):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for the box (0, in_size): (bounds [out,2] int32 = first input
    index and tap count, coefficients [out, ksize] int32 in 22-bit fixed point, ksize)."""
    f, support = FILTERS[filt]
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = support * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        xmin = max(xmin, 0)
        xmax = int(center + support + 0.5)
        xmax = min(xmax, in_size) - xmin
        w = [f((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = sum(w)
        if ww != 0.0:
            w = [v / ww for v in w]
        for x, v in enumerate(w):
            kk[xx, x] = int(v * (1 << PRECISION_BITS) + (0.5 if v >= 0 else -0.5))     # C truncation toward zero
        bounds[xx] = (xmin, xmax)
    return bounds, kk, ksize


def _pass(src, bounds, kk, axis):
    """One 8bpc pass along ``axis`` of src [H,W,C] uint8: ss = 2^21 + sum pixel * k; clip8(ss >> 22)."""
    src = np.moveaxis(src.astype(np.int64), axis, 0)
    out = np.empty((bounds.shape[0],) + src.shape[1:], np.int64)
    for xx in range(bounds.shape[0]):
        x0, n = int(bounds[xx, 0]), int(bounds[xx, 1])
        acc = np.full(src.shape[1:], 1 << (PRECISION_BITS - 1), np.int64)
        for x in range(n):
            acc += src[x0 + x] * int(kk[xx, x])
        out[xx] = acc >> PRECISION_BITS
    return np.moveaxis(np.clip(out, 0, 255).astype(np.uint8), 0, axis)


def resample_u8(img, new_h, new_w, filt='lanczos'):
    """img uint8 [H,W,C] -> [new_h,new_w,C]; the horizontal pass first (only when the width changes), 8-bit in between."""
    h, w = img.shape[:2]
    out = img
    if new_w != w:
        b, k, _ = precompute_coeffs(w, new_w, filt)
        out = _pass(out, b, k, 1)
    if new_h != h:
        b, k, _ = precompute_coeffs(h, new_h, filt)
        out = _pass(out, b, k, 0)
    return out


def resize_and_padding_mask_image(img_u8, new_size, margin_size=10):
    """img_u8 [H,W,3] uint8 (the RGB sketch) -> [new_size,new_size,3] uint8 (input_pipeline.py:199-239)."""
    h, w = img_u8.shape[:2]
    scale = new_size / max(h + 2 * margin_size, w + 2 * margin_size)
    new_h, new_w = int(round(h * scale)), int(round(w * scale))
    assert new_h <= new_size and new_w <= new_size
    plane = img_u8[:, :, :1]
    if scale != 1:
        plane = resample_u8(plane, new_h, new_w, 'lanczos')
    top, left = (new_size - new_h) // 2, (new_size - new_w) // 2
    canvas = np.full((new_size, new_size), 255, np.uint8)
    canvas[top:top + new_h, left:left + new_w] = plane[:, :, 0]
    return np.repeat(canvas[:, :, None], 3, axis=2)


def reverse_resize_image(inst_u8, box_h, box_w, h_w_ratio=1, margin_size=10):
    """[S,S,3] uint8 generated instance -> [box_h, box_w, 3]: cut the padding, bilinear resize to the box plus margins,
    cut the margins (Pipeline_utils/fg_color_utils.py:137-160; scipy.misc.imresize = PIL resize, bilinear)."""
    s = inst_u8.shape[0]
    bh, bw = box_h + 2 * margin_size, box_w + 2 * margin_size
    if bh * h_w_ratio > bw:
        pad = int(round(s * (bh * h_w_ratio - bw) / (bh * h_w_ratio) / 2.))
        cut = inst_u8[:, pad:s - pad]
    else:
        pad = int(round(s * (bw - bh * h_w_ratio) / bw / 2.))
        cut = inst_u8[pad:s - pad, :]
    rev = resample_u8(np.ascontiguousarray(cut), bh, bw, 'bilinear')
    return rev[margin_size:margin_size + box_h, margin_size:margin_size + box_w]


def thicken_drawings(image):
    """2x2 grey dilation of the dark strokes: neighbourhood rows {i,i+1} x cols {j,j+1} (skimage pads an even footprint
    at the start), replicated to 3 channels."""
    img = 255 - np.array(image[:, :, 0], dtype=np.uint8)
    p = np.pad(img, ((0, 1), (0, 1)), mode='edge')
    dil = np.maximum(np.maximum(p[:-1, :-1], p[1:, :-1]), np.maximum(p[:-1, 1:], p[1:, 1:]))
    return np.repeat((255 - dil)[:, :, None], 3, axis=2).astype(np.uint8)


def sketch_preprocess(sk_u8, thicken=False):
    """uint8 [N,H,W,3] -> float32 [N,H,W,3] in [-1,1] (main_procedure.py:577-583), optionally thickened first."""
    if thicken:
        sk_u8 = np.stack([thicken_drawings(s) for s in sk_u8])
    return (sk_u8.astype(np.float32) / np.float32(255.) * np.float32(2.) - np.float32(1)).astype(np.float32)


def image_postprocess(x):
    """float [..] in [-1,1] -> uint8 with the reference's truncating cast (main_procedure.py:601-610)."""
    return (((np.asarray(x, np.float32) + 1) / 2.) * 255).astype(np.uint8)


def decode_paired_example(img_u8, sk_u8, size, noise=None, distance_map=False):
    """Raw 384x384x3 record images -> (image, sketch) float32 NCHW in [-1,1] as get_paired_input does
    (input_pipeline.py:77-131): TF1 bilinear at an integer factor = the source pixel at (f*y, f*x); TF1 area = mean of the
    f x f block; (v - min)/(max - min + 1); + dequantisation noise; *2-1."""
    r = img_u8.shape[0]
    f = r // size
    assert f * size == r
    img = img_u8.astype(np.float32)
    sk = sk_u8.astype(np.float32)
    if distance_map:
        from scipy import ndimage
        sk = np.where(sk < 250, 0.0, 255.0).astype(np.float32)
        sk = ndimage.distance_transform_edt(sk).astype(np.float32)
        sk = sk / sk.max() * 255.0
    if f != 1:
        img = img[::f, ::f]
        sk = sk.reshape(size, f, size, f, 3).mean(axis=(1, 3))
    img = (img - img.min()) / (img.max() - img.min() + 1)
    if noise is not None:
        img = img + noise
    img = img * 2.0 - 1.0
    sk = sk / 255.0 * 2.0 - 1.0
    return (np.ascontiguousarray(img.transpose(2, 0, 1), dtype=np.float32),
            np.ascontiguousarray(sk.transpose(2, 0, 1), dtype=np.float32))
