"""Goldens for the Background_Colorization host image loading, produced by the REFERENCE module itself:
/root/reference/Background_Colorization/data_processing/image_processing.py imports without TensorFlow (numpy + PIL only), so
load_image / load_region_mask are run on small png files written here and their outputs stored.  Only possible in the build
container (the reference does not travel); run from the repo root:  python tests/golden/make_bg_image_goldens.py"""
import importlib.util
import os

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference/Background_Colorization/data_processing/image_processing.py'
spec = importlib.util.spec_from_file_location('ref_image_processing', REF)
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

rng = np.random.RandomState(20260928)
d = os.path.join(HERE, 'bg_images')
os.makedirs(d, exist_ok=True)
out = {}
# images: one already at the target size, one to be resized (non-square, odd sizes), one grey-scale file, one RGBA file
files = {'same.png': rng.randint(0, 256, (24, 24, 3)).astype(np.uint8),
         'wide.png': rng.randint(0, 256, (17, 41, 3)).astype(np.uint8),
         'grey.png': rng.randint(0, 256, (30, 20)).astype(np.uint8),
         'rgba.png': rng.randint(0, 256, (9, 13, 4)).astype(np.uint8)}
for name, a in files.items():
    Image.fromarray(a).save(os.path.join(d, name))
    out['image/' + name] = ref.load_image(os.path.join(d, name), 24)
# segment maps: the three label values plus values that must map to 0; a palette-less RGB file whose channels differ
seg = np.zeros((24, 24, 3), np.uint8)
seg[:10] = 128
seg[10:18] = 255
seg[18:, :8] = 127
seg[18:, 8:16] = 129
seg[18:, 16:] = 254
seg[2:5, 3:9, 0] = 7            # red channel decides
seg[6:8, 3:9, 1:] = 255         # green / blue are ignored
Image.fromarray(seg).save(os.path.join(d, 'seg.png'))
out['mask/seg.png'] = ref.load_region_mask(os.path.join(d, 'seg.png'), 24, False)
out['mask/seg.png/test'] = ref.load_region_mask(os.path.join(d, 'seg.png'), 24, True)
big = np.full((31, 19, 3), 255, np.uint8)
big[:, :7] = 128
Image.fromarray(big).save(os.path.join(d, 'seg_other_size.png'))
out['mask/seg_other_size.png'] = ref.load_region_mask(os.path.join(d, 'seg_other_size.png'), 24, False)     # NOT resized
np.savez_compressed(os.path.join(HERE, 'bg_image_goldens.npz'), **out)
print({k: (v.shape, str(v.dtype)) for k, v in out.items()})
