"""Host-side helpers of obj_lib/input_pipeline.py that the hot path's callers use
(:11-15 num_classes, :184-196 split_inputs, :199-257 sketch pre-processing).  The TFRecord queue
machinery is out of scope: training data comes from
sketchyscenecolorization_amd.synthetic or from the caller."""
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
