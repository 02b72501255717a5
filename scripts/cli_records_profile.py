"""cProfile of the training loop's own thread while it trains from records (where does an iteration's host time go?)"""
import cProfile
import os
import pstats
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np                                                  # noqa: E402
from sketchyscenecolorization_amd import tfrecord as tf             # noqa: E402

d = tempfile.mkdtemp()
os.makedirs(os.path.join(d, 'data', 'tfrecord', 'train'))
rng = np.random.RandomState(0)
for f in range(4):
    recs = []
    for i in range(48):
        sk = np.full((384, 384, 3), 255, np.uint8)
        sk[(7 * i) % 370:(7 * i) % 370 + 6, 40:340] = 0
        text = np.zeros(15, np.uint8)
        text[-4:] = rng.randint(2, 58, 4)
        recs.append(tf.make_example({'ImageName': b'x.png', 'cartoon_data': rng.randint(0, 256, (384, 384, 3)).astype(np.uint8).tobytes(),
                                     'sketch_data': sk.tobytes(), 'Category': b'car', 'Category_id': i % 25,
                                     'Color_text': b'the car is red', 'Text_vocab_indices': text.tobytes()}))
    tf.write_records(os.path.join(d, 'data', 'tfrecord', 'train', '%d.tfrecord' % f), recs)
os.chdir(d)
import obj_colorization_main as cli                                 # noqa: E402
pr = cProfile.Profile()
pr.enable()
cli.main(['--mode', 'train', '-bt', 'Pix2Pix', '-si', '0', '-bs', '32', '-mi', '260', '-smf', '100000', '-swf', '100', '-clt', '100'])
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(30)
