"""obj_colorization_main.py --mode train on data/tfrecord/train (a synthetic dataset in the reference's record format, written
here) vs the synthetic queue: seconds per iteration.  usage: cli_train_rate_records.py [records] [batch]"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np                                                  # noqa: E402
from sketchyscenecolorization_amd import tfrecord as tf             # noqa: E402

n, bs = int(sys.argv[1]) if len(sys.argv) > 1 else 192, int(sys.argv[2]) if len(sys.argv) > 2 else 32
d = tempfile.mkdtemp()
os.makedirs(os.path.join(d, 'data', 'tfrecord', 'train'))
rng = np.random.RandomState(0)
for f in range(4):
    recs = []
    for i in range(n // 4):
        sk = np.full((384, 384, 3), 255, np.uint8)
        sk[(7 * i) % 370:(7 * i) % 370 + 6, 40:340] = 0
        text = np.zeros(15, np.uint8)
        ln = rng.randint(2, 11) if os.environ.get('RECORDS_VARY_CAPTIONS') == '1' else 4     # left-padded captions of 2..10 tokens
        text[15 - ln:] = rng.randint(2, 58, ln)
        recs.append(tf.make_example({'ImageName': b'x.png', 'cartoon_data': rng.randint(0, 256, (384, 384, 3)).astype(np.uint8).tobytes(),
                                     'sketch_data': sk.tobytes(), 'Category': b'car', 'Category_id': i % 25,
                                     'Color_text': b'the car is red', 'Text_vocab_indices': text.tobytes()}))
    tf.write_records(os.path.join(d, 'data', 'tfrecord', 'train', '%d.tfrecord' % f), recs)
for lazy, pre, gr in (('1', '1', '1'), ('1', '1', '0'), ('1', '0', '1'), ('0', '0', '0'), ('1', '1', '1')):
    env = dict(os.environ, SSC_CLI_LAZY_LOSS=lazy, SSC_RECORD_PREFETCH=pre, SSC_TRAIN_GRAPHS=gr)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'obj_colorization_main.py'), '--mode', 'train', '-bt', 'Pix2Pix', '-si', '0',
                          '-bs', str(bs), '-mi', '400', '-smf', '100000', '-swf', '100', '-clt', '100'], cwd=d, env=env,
                         capture_output=True, text=True)
    ts = [l.split('Average time: ')[1] for l in out.stdout.splitlines() if 'Average time' in l and 'inf' not in l]
    print('[records, lazy losses %s, prefetch %s, graphs %s]' % (lazy, pre, gr), ' '.join(ts) if ts else out.stderr[-1500:])
