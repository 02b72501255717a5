"""Host-side image loading of the Background_Colorization module (reference Background_Colorization/data_processing/
image_processing.py:5-25): same names, arguments and results -- pinned by goldens produced by the reference module itself
(tests/golden/make_bg_image_goldens.py)."""
import numpy as np


def load_image(imname, image_size):
    """RGB uint8 [1, image_size, image_size, 3]; any other size is resized with PIL's BILINEAR filter (:5-11)."""
    from PIL import Image
    im = Image.open(imname).convert('RGB')
    if im.width != image_size or im.height != image_size:
        im = im.resize((image_size, image_size), resample=Image.BILINEAR)
    return np.array(im, dtype=np.uint8)[None]


def load_region_mask(seg_path, image_size, is_test=False):
    """int32 [1, H, W] region labels from the red channel of the segment png: 128 -> 1 (sky), 255 -> 2 (ground), anything else
    0 (foreground); all zeros in test mode, where no segment image exists (:14-25).  The png is NOT resized."""
    if is_test:
        return np.zeros([1, image_size, image_size], dtype=np.int32)
    from PIL import Image
    seg = np.array(Image.open(seg_path).convert('RGB'), dtype=np.uint8)[:, :, 0]
    lab = np.zeros(seg.shape, dtype=np.int32)
    lab[seg == 128] = 1
    lab[seg == 255] = 2
    return lab[None]
