"""TensorFlow V2 checkpoints ("tensor bundles", what tf.train.Saver writes and the authors' released models are)
without TensorFlow.

    <prefix>.index                 an SSTable (LevelDB table format) : tensor name -> BundleEntryProto,
                                   plus the entry "" -> BundleHeaderProto
    <prefix>.data-0000k-of-0000n   the raw little-endian tensor bytes, addressed by (shard_id, offset, size)

Restated from the published formats (LevelDB ``table_format.md``; tensorflow/core/protobuf/tensor_bundle.proto):
  * table file = data blocks | metaindex block | index block | 48-byte footer (two BlockHandles as varint64 pairs,
    zero padding, magic 0xdb4775248b80fb57 little-endian);
  * block = entries (varint32 shared, varint32 unshared, varint32 value_len, key suffix, value) | uint32 restarts[] |
    uint32 num_restarts, followed in the file by a 5-byte trailer (compression type, masked CRC-32C);
  * BundleEntryProto {dtype=1, shape=2 {dim=2 {size=1}}, shard_id=3, offset=4, size=5, crc32c=6 (fixed32, masked)}.
No TensorFlow and no released checkpoint is available in this environment, so this reader is checked against the
writer below (same specification) and against hand-assembled byte strings in tests/test_tf_checkpoint.py -- it has NOT
been run on a file written by TensorFlow itself.
"""
import os
import struct

import numpy as np

from . import tfrecord
from .tfrecord import _enc_varint, _fields, _ld, _varint

MAGIC = 0xdb4775248b80fb57
DTYPES = {1: np.dtype('<f4'), 2: np.dtype('<f8'), 3: np.dtype('<i4'), 9: np.dtype('<i8'), 4: np.dtype('u1'),
          10: np.dtype('bool')}
DT_OF = {np.dtype('float32'): 1, np.dtype('float64'): 2, np.dtype('int32'): 3, np.dtype('int64'): 9,
         np.dtype('uint8'): 4, np.dtype('bool'): 10}


# ------------------------------------------------------------------ SSTable
def _block(buf, offset, size, verify=True):
    data = buf[offset:offset + size]
    ctype = buf[offset + size]
    if verify:
        want = struct.unpack('<I', buf[offset + size + 1:offset + size + 5])[0]
        if tfrecord.masked_crc(bytes(buf[offset:offset + size + 1])) != want:
            raise IOError('corrupt table block at %d' % offset)
    if ctype != 0:
        raise NotImplementedError('compressed table blocks (type %d): TensorFlow writes bundle indexes uncompressed'
                                  % ctype)
    return data


def _entries(block):
    n_restarts = struct.unpack('<I', block[-4:])[0]
    limit = len(block) - 4 - 4 * n_restarts
    pos, key = 0, b''
    mv = memoryview(block)
    while pos < limit:
        shared, pos = _varint(mv, pos)
        unshared, pos = _varint(mv, pos)
        vlen, pos = _varint(mv, pos)
        key = key[:shared] + bytes(mv[pos:pos + unshared])
        pos += unshared
        yield key, bytes(mv[pos:pos + vlen])
        pos += vlen


def _handle(buf, pos=0):
    off, pos = _varint(memoryview(buf), pos)
    size, pos = _varint(memoryview(buf), pos)
    return off, size, pos


def read_table(path, verify=True):
    """{key bytes: value bytes} of an SSTable file."""
    buf = open(path, 'rb').read()
    if len(buf) < 48 or struct.unpack('<Q', buf[-8:])[0] != MAGIC:
        raise IOError('%s: not a TensorFlow/LevelDB table (bad magic)' % path)
    footer = buf[-48:]
    _mo, _ms, p = _handle(footer, 0)
    io_, is_, _ = _handle(footer, p)
    out = {}
    for _sep, hv in _entries(_block(buf, io_, is_, verify)):
        bo, bs, _ = _handle(hv)
        for k, v in _entries(_block(buf, bo, bs, verify)):
            out[k] = v
    return out


def _build_block(items, restart_interval=16):
    out, restarts, prev = bytearray(), [], b''
    for i, (k, v) in enumerate(items):
        if i % restart_interval == 0:
            restarts.append(len(out))
            shared = 0
        else:
            shared = 0
            while shared < min(len(prev), len(k)) and prev[shared] == k[shared]:
                shared += 1
        out += _enc_varint(shared) + _enc_varint(len(k) - shared) + _enc_varint(len(v)) + k[shared:] + v
        prev = k
    if not restarts:
        restarts = [0]
    for r in restarts:
        out += struct.pack('<I', r)
    out += struct.pack('<I', len(restarts))
    return bytes(out)


def write_table(path, mapping, block_bytes=4096):
    """Write {key bytes: value bytes} as an uncompressed SSTable (keys in bytewise order)."""
    f = bytearray()

    def emit(block):
        off = len(f)
        f.extend(block)
        f.append(0)
        f.extend(struct.pack('<I', tfrecord.masked_crc(block + b'\x00')))
        return off, len(block)

    index, cur, cur_size = [], [], 0
    items = sorted(mapping.items())
    for k, v in items:
        cur.append((k, v))
        cur_size += len(k) + len(v) + 8
        if cur_size >= block_bytes:
            index.append((cur[-1][0], emit(_build_block(cur))))
            cur, cur_size = [], 0
    if cur or not index:
        index.append((cur[-1][0] if cur else b'', emit(_build_block(cur))))
    meta = emit(_build_block([]))
    idx = emit(_build_block([(k, _enc_varint(o) + _enc_varint(s)) for k, (o, s) in index], restart_interval=1))
    footer = _enc_varint(meta[0]) + _enc_varint(meta[1]) + _enc_varint(idx[0]) + _enc_varint(idx[1])
    footer += b'\x00' * (40 - len(footer)) + struct.pack('<Q', MAGIC)
    f.extend(footer)
    with open(path, 'wb') as fp:
        fp.write(bytes(f))


# ------------------------------------------------------------------ bundle
def _parse_entry(buf):
    e = {'dtype': 0, 'shape': [], 'shard_id': 0, 'offset': 0, 'size': 0, 'crc32c': None, 'sliced': False}
    for num, wt, val in _fields(memoryview(buf)):
        if num == 1:
            e['dtype'] = val
        elif num == 2:
            for n2, _w2, dim in _fields(val):
                if n2 == 2:
                    size = 0
                    for n3, _w3, v3 in _fields(dim):
                        if n3 == 1:
                            size = v3
                    e['shape'].append(size)
        elif num == 3:
            e['shard_id'] = val
        elif num == 4:
            e['offset'] = val
        elif num == 5:
            e['size'] = val
        elif num == 6:
            e['crc32c'] = struct.unpack('<I', bytes(val))[0]
        elif num == 7:
            e['sliced'] = True
    return e


def list_variables(prefix):
    """[(name, shape, numpy dtype)] of a checkpoint prefix (``.../model_100.ckpt-100``)."""
    out = []
    for k, v in sorted(read_table(prefix + '.index').items()):
        if k == b'':
            continue
        e = _parse_entry(v)
        out.append((k.decode('utf-8'), tuple(e['shape']), DTYPES.get(e['dtype'])))
    return out


def read_checkpoint(prefix, names=None, verify=True):
    """{variable name: ndarray} for every (or the requested) float / int variable of a V2 checkpoint."""
    table = read_table(prefix + '.index', verify)
    num_shards = 1
    for num, _wt, val in _fields(memoryview(table.get(b'', b''))):
        if num == 1:
            num_shards = val
    shards = {}
    out = {}
    for k, v in table.items():
        if k == b'':
            continue
        name = k.decode('utf-8')
        if names is not None and name not in names:
            continue
        e = _parse_entry(v)
        if e['sliced'] or e['dtype'] not in DTYPES:
            continue            # partitioned variables / strings: nothing on this path uses them
        sid = e['shard_id']
        if sid not in shards:
            shards[sid] = open('%s.data-%05d-of-%05d' % (prefix, sid, num_shards), 'rb')
        fh = shards[sid]
        fh.seek(e['offset'])
        raw = fh.read(e['size'])
        if len(raw) != e['size']:
            raise IOError('%s: truncated data for %s' % (prefix, name))
        if verify and e['crc32c'] is not None and tfrecord.masked_crc(raw) != e['crc32c']:
            raise IOError('%s: checksum mismatch for %s' % (prefix, name))
        out[name] = np.frombuffer(raw, dtype=DTYPES[e['dtype']]).reshape(e['shape']).copy()
    for fh in shards.values():
        fh.close()
    return out


def write_checkpoint(prefix, tensors):
    """Write {name: ndarray} as a single-shard V2 checkpoint that tf.train.Saver / tf.train.load_checkpoint can read."""
    table = {b'': _enc_varint((1 << 3) | 0) + _enc_varint(1) +                # num_shards = 1
             _ld(3, _enc_varint((1 << 3) | 0) + _enc_varint(1))}               # version { producer: 1 }
    offset = 0
    with open('%s.data-00000-of-00001' % prefix, 'wb') as data:
        for name in sorted(tensors):
            a = np.asarray(tensors[name])
            if a.dtype not in DT_OF:
                a = a.astype(np.float32)
            raw = a.astype(a.dtype.newbyteorder('<')).tobytes(order='C')
            shape = b''.join(_ld(2, _enc_varint((1 << 3) | 0) + _enc_varint(int(d))) for d in a.shape)
            entry = _enc_varint((1 << 3) | 0) + _enc_varint(DT_OF[a.dtype]) + _ld(2, shape)
            entry += _enc_varint((4 << 3) | 0) + _enc_varint(offset) + _enc_varint((5 << 3) | 0) + _enc_varint(len(raw))
            entry += _enc_varint((6 << 3) | 5) + struct.pack('<I', tfrecord.masked_crc(raw))
            table[name.encode('utf-8')] = entry
            data.write(raw)
            offset += len(raw)
    write_table(prefix + '.index', table)


def is_tf_checkpoint(prefix):
    return os.path.exists(prefix + '.index')
