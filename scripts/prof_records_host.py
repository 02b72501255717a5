import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sketchyscenecolorization_amd import tfrecord as tf
d = tempfile.mkdtemp()
rng = np.random.RandomState(0)
recs = []
for i in range(64):
    sk = np.full((384, 384, 3), 255, np.uint8)
    text = np.zeros(15, np.uint8)
    recs.append(tf.make_example({'ImageName': b'x.png', 'cartoon_data': rng.randint(0, 256, (384, 384, 3)).astype(np.uint8).tobytes(),
                                 'sketch_data': sk.tobytes(), 'Category': b'car', 'Category_id': i % 25,
                                 'Color_text': b'the car is red', 'Text_vocab_indices': text.tobytes()}))
p = os.path.join(d, 'a.tfrecord')
tf.write_records(p, recs)
for verify in (True, False):
    t = time.time(); rs = list(tf.read_records(p, verify=verify)); dt = time.time() - t
    print('read_records verify=%s: %.3f ms/record' % (verify, dt / 64 * 1e3))
t = time.time(); fs = [tf.parse_example(r) for r in rs]; print('parse_example: %.3f ms/record' % ((time.time() - t) / 64 * 1e3))
t = time.time()
raw = np.empty((2, 64, 384, 384, 3), np.uint8)
for k, f in enumerate(fs):
    raw[0, k] = np.frombuffer(f['cartoon_data'][0], dtype=np.uint8).reshape(384, 384, 3)
    raw[1, k] = np.frombuffer(f['sketch_data'][0], dtype=np.uint8).reshape(384, 384, 3)
print('frombuffer copies: %.3f ms/record' % ((time.time() - t) / 64 * 1e3))
import cProfile, pstats
cProfile.run('list(map(tf.parse_example, rs))', '/tmp/pe.prof')
pstats.Stats('/tmp/pe.prof').sort_stats('cumtime').print_stats(8)
