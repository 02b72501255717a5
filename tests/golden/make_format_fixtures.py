"""Byte-level fixtures of the two TensorFlow file formats the readers parse, assembled here from the published format
descriptions with this script's OWN varint / protobuf / CRC-32C / LevelDB-table code (nothing from the package under
test, no TensorFlow):

  fixture.tfrecord                 TFRecord framing: uint64 length, uint32 masked crc32c(length), payload, uint32 masked
                                   crc32c(payload)  (tensorflow/core/lib/io/record_writer.cc) around tf.train.Example
                                   messages (tensorflow/core/example/{example,feature}.proto) with the feature names of
                                   data_preparation.py:13-96
  fixture_ckpt.index / .data-00000-of-00001
                                   tensor bundle: a LevelDB SSTable (leveldb doc/table_format.md: prefix-compressed entries,
                                   restart array, 5-byte block trailer with masked crc32c, 48-byte footer with magic
                                   0xdb4775248b80fb57) mapping "" -> BundleHeaderProto and variable name -> BundleEntryProto
                                   (tensorflow/core/protobuf/tensor_bundle.proto), tensor bytes in the data shard

run from the repo root:  python tests/golden/make_format_fixtures.py"""
import os
import struct

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def crc32c(data):                       # bitwise CRC-32C (Castagnoli, reflected polynomial 0x82F63B78)
    crc = 0xFFFFFFFF
    for b in data:
        crc ^= b
        for _ in range(8):
            crc = (crc >> 1) ^ (0x82F63B78 if crc & 1 else 0)
    return crc ^ 0xFFFFFFFF


def masked(crc):
    return ((((crc >> 15) | (crc << 17)) & 0xFFFFFFFF) + 0xa282ead8) & 0xFFFFFFFF


def varint(n):
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def field(num, wire, payload):          # wire 0: varint value, 2: length-delimited bytes, 5: fixed32
    key = varint((num << 3) | wire)
    if wire == 0:
        return key + varint(payload)
    if wire == 2:
        return key + varint(len(payload)) + payload
    return key + struct.pack('<I', payload)


# ------------------------------------------------------------------ TFRecord of tf.train.Example
def bytes_feature(v):
    return field(1, 2, field(1, 2, v))                          # Feature.bytes_list(1) { value(1) }


def int64_feature(v):
    return field(3, 2, field(1, 2, varint(v)))                  # Feature.int64_list(3) { packed value(1) }


def example(feats):
    body = b''
    for k, v in feats:                                          # Features.feature(1): map<string, Feature> entries
        body += field(1, 2, field(1, 2, k.encode()) + field(2, 2, v))
    return field(1, 2, body)                                    # Example.features(1)


def record(payload):
    n = struct.pack('<Q', len(payload))
    return n + struct.pack('<I', masked(crc32c(n))) + payload + struct.pack('<I', masked(crc32c(payload)))


ex0 = example([('ImageName', bytes_feature(b'L0_sample7_1.png')), ('cartoon_data', bytes_feature(bytes(range(48)))),
               ('sketch_data', bytes_feature(bytes([255] * 40 + [0] * 8))), ('Category', bytes_feature(b'car')),
               ('Category_id', int64_feature(4)), ('Color_text', bytes_feature(b'the car is red with black windows')),
               ('Text_vocab_indices', bytes_feature(bytes([0] * 8 + [3, 9, 4, 21, 5, 7, 30])))])
ex1 = example([('ImageName', bytes_feature(b'L0_sample9_3.png')), ('cartoon_data', bytes_feature(bytes(range(200, 248)))),
               ('sketch_data', bytes_feature(bytes([0] * 48))), ('Category', bytes_feature(b'tree')),
               ('Category_id', int64_feature(23)), ('Color_text', bytes_feature(b'the tree is green')),
               ('Text_vocab_indices', bytes_feature(bytes([0] * 11 + [3, 40, 4, 12])))])
with open(os.path.join(HERE, 'fixture.tfrecord'), 'wb') as f:
    f.write(record(ex0) + record(ex1))

# ------------------------------------------------------------------ tensor bundle
DT_FLOAT, DT_INT64 = 1, 9
tensors = [('discriminator/Conv/prelu/param', np.float32(0.2).reshape(())),
           ('generator/Conv/biases', np.arange(4, dtype=np.float32).reshape(1, 4, 1, 1) - 1.5),
           ('generator/encoder_1/conv/filter', (np.arange(24, dtype=np.float32).reshape(2, 2, 2, 3) * 0.25 - 2.0)),
           ('generator/encoder_1/conv/filter/Adam_1', np.full((2, 2, 2, 3), 0.5, np.float32)),
           ('global_step', np.array(1234, np.int64))]
data = b''
entries = []
for name, arr in tensors:
    raw = np.ascontiguousarray(arr).tobytes()
    shape = b''.join(field(2, 2, field(1, 0, int(d))) for d in arr.shape)      # TensorShapeProto.dim(2) { size(1) }
    e = field(1, 0, DT_INT64 if arr.dtype == np.int64 else DT_FLOAT) + field(2, 2, shape)
    if len(data):
        e += field(4, 0, len(data))                                             # offset (proto3: zero is omitted)
    e += field(5, 0, len(raw)) + field(6, 5, masked(crc32c(raw)))               # size, crc32c (fixed32, masked)
    entries.append((name.encode(), e))
    data += raw
header = field(1, 0, 1) + field(3, 2, field(1, 0, 1))                           # num_shards = 1, version { producer = 1 }
entries = [(b'', header)] + sorted(entries)


def block(kvs, restart_interval):
    out, restarts, last = b'', [], b''
    for i, (k, v) in enumerate(kvs):
        shared = 0
        if i % restart_interval == 0:
            restarts.append(len(out))
        else:
            while shared < min(len(k), len(last)) and k[shared] == last[shared]:
                shared += 1
        out += varint(shared) + varint(len(k) - shared) + varint(len(v)) + k[shared:] + v
        last = k
    out += b''.join(struct.pack('<I', r) for r in restarts) + struct.pack('<I', len(restarts))
    return out


def with_trailer(body):
    return body + b'\x00' + struct.pack('<I', masked(crc32c(body + b'\x00')))   # type 0 = no compression


table = b''
index_entries = []
for chunk in (entries[:3], entries[3:]):                                        # two data blocks
    body = block(chunk, restart_interval=2)
    index_entries.append((chunk[-1][0], varint(len(table)) + varint(len(body))))
    table += with_trailer(body)
meta_body = block([], 1)
meta_handle = varint(len(table)) + varint(len(meta_body))
table += with_trailer(meta_body)
index_body = block(index_entries, 1)
index_handle = varint(len(table)) + varint(len(index_body))
table += with_trailer(index_body)
footer = meta_handle + index_handle
table += footer + b'\x00' * (40 - len(footer)) + struct.pack('<Q', 0xdb4775248b80fb57)
with open(os.path.join(HERE, 'fixture_ckpt.index'), 'wb') as f:
    f.write(table)
with open(os.path.join(HERE, 'fixture_ckpt.data-00000-of-00001'), 'wb') as f:
    f.write(data)
print('wrote fixture.tfrecord (%d bytes), fixture_ckpt.index (%d), fixture_ckpt.data (%d)' % (len(record(ex0) + record(ex1)), len(table), len(data)))
