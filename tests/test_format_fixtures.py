"""The TFRecord / tf.train.Example reader and the tensor-bundle checkpoint reader on byte-level fixtures assembled from
the published format descriptions by tests/golden/make_format_fixtures.py (its own varint, protobuf, CRC-32C and table
code; nothing from the package, no TensorFlow)."""
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(__file__), 'golden')


def test_tfrecord_reader_on_the_hand_assembled_file():
    from sketchyscenecolorization_amd import tfrecord as R
    recs = list(R.read_records(os.path.join(G, 'fixture.tfrecord')))
    assert len(recs) == 2
    a, b = R.parse_example(recs[0]), R.parse_example(recs[1])
    assert a['ImageName'] == [b'L0_sample7_1.png'] and a['Category'] == [b'car'] and a['Category_id'] == [4]
    assert a['cartoon_data'] == [bytes(range(48))] and a['sketch_data'] == [bytes([255] * 40 + [0] * 8)]
    assert a['Text_vocab_indices'] == [bytes([0] * 8 + [3, 9, 4, 21, 5, 7, 30])]
    assert a['Color_text'] == [b'the car is red with black windows']
    assert b['Category'] == [b'tree'] and b['Category_id'] == [23] and b['cartoon_data'] == [bytes(range(200, 248))]
    assert np.frombuffer(b['Text_vocab_indices'][0], np.uint8).tolist() == [0] * 11 + [3, 40, 4, 12]


def test_tfrecord_reader_rejects_a_corrupted_payload(tmp_path):
    from sketchyscenecolorization_amd import tfrecord as R
    raw = bytearray(open(os.path.join(G, 'fixture.tfrecord'), 'rb').read())
    raw[40] ^= 0x01
    p = os.path.join(tmp_path, 'bad.tfrecord')
    open(p, 'wb').write(bytes(raw))
    with pytest.raises(IOError):
        list(R.read_records(p))


def test_checkpoint_reader_on_the_hand_assembled_bundle():
    from sketchyscenecolorization_amd import tf_checkpoint as C
    pre = os.path.join(G, 'fixture_ckpt')
    assert C.is_tf_checkpoint(pre)
    t = C.read_checkpoint(pre)
    assert sorted(t) == ['discriminator/Conv/prelu/param', 'generator/Conv/biases', 'generator/encoder_1/conv/filter',
                         'generator/encoder_1/conv/filter/Adam_1', 'global_step']
    assert t['discriminator/Conv/prelu/param'].shape == () and t['discriminator/Conv/prelu/param'] == np.float32(0.2)
    assert np.array_equal(t['generator/Conv/biases'], np.arange(4, dtype=np.float32).reshape(1, 4, 1, 1) - 1.5)
    assert np.array_equal(t['generator/encoder_1/conv/filter'], np.arange(24, dtype=np.float32).reshape(2, 2, 2, 3) * 0.25 - 2.0)
    assert np.array_equal(t['generator/encoder_1/conv/filter/Adam_1'], np.full((2, 2, 2, 3), 0.5, np.float32))
    assert t['global_step'].dtype == np.int64 and int(t['global_step']) == 1234
    assert ('global_step', (), np.dtype('<i8')) in C.list_variables(pre)
