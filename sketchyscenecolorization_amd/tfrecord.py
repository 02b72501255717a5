"""TFRecord files and tf.train.Example records without TensorFlow.

The reference feeds training from ``data/tfrecord/<mode>/*.tfrecord`` through ``tf.TFRecordReader`` +
``tf.parse_single_example`` (obj_lib/input_pipeline.py:57-72).  This module reads (and, for tests, writes) the same
container:

    record  := uint64 length | uint32 masked_crc32c(length) | byte data[length] | uint32 masked_crc32c(data)
    masked  := ((crc >> 15) | (crc << 17)) + 0xa282ead8   (mod 2^32)

and decodes the ``Example`` protobuf (Example{features=1: Features{feature=1: map<string, Feature>}}, Feature = oneof
{bytes_list=1, float_list=2, int64_list=3}, each a message with repeated ``value = 1``) with a minimal wire-format
parser.  CRCs are checked with the C-ABI host function ``ssc_crc32c``.
"""
import ctypes
import os
import struct

import numpy as np

_MASK_DELTA = 0xa282ead8


def crc32c(data):
    """CRC-32C of a bytes-like object, read in place (a record of the reference's dataset is 884 KB: no copy for the call)."""
    from . import hip
    fn = hip.lib().ssc_crc32c
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int64]
    fn.restype = ctypes.c_uint32
    a = np.frombuffer(data, dtype=np.uint8)
    return int(fn(a.ctypes.data if a.size else None, a.size))


def masked_crc(data):
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + _MASK_DELTA) & 0xffffffff


# ------------------------------------------------------------------ container
def read_records(path, verify=True, views=False):
    """Yield the payload of every record of a .tfrecord file.  views: the file is mapped and the payloads are memoryview
    slices of the mapping (they keep it alive): nothing is copied until somebody copies it -- the training queue checks the
    CRC in place and copies the two images of a record once, into its pinned staging buffer."""
    if views:
        import mmap
        with open(path, 'rb') as f:
            size = os.fstat(f.fileno()).st_size
            if size == 0:
                return
            mm = memoryview(mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ))
        pos = 0
        while pos < size:
            if pos + 12 > size:
                raise IOError('%s: truncated record header' % path)
            length, len_crc = struct.unpack_from('<QI', mm, pos)
            if verify and masked_crc(mm[pos:pos + 8]) != len_crc:
                raise IOError('%s: corrupt record length' % path)
            if pos + 12 + length + 4 > size:
                raise IOError('%s: truncated record' % path)
            data = mm[pos + 12:pos + 12 + length]
            if verify and masked_crc(data) != struct.unpack_from('<I', mm, pos + 12 + length)[0]:
                raise IOError('%s: corrupt record data' % path)
            yield data
            pos += 16 + length
        return
    with open(path, 'rb') as f:
        while True:
            head = f.read(12)
            if not head:
                return
            if len(head) != 12:
                raise IOError('%s: truncated record header' % path)
            length, len_crc = struct.unpack('<QI', head)
            if verify and masked_crc(head[:8]) != len_crc:
                raise IOError('%s: corrupt record length' % path)
            data = f.read(length)
            tail = f.read(4)
            if len(data) != length or len(tail) != 4:
                raise IOError('%s: truncated record' % path)
            if verify and masked_crc(data) != struct.unpack('<I', tail)[0]:
                raise IOError('%s: corrupt record data' % path)
            yield data


def write_records(path, payloads):
    with open(path, 'wb') as f:
        for data in payloads:
            head = struct.pack('<Q', len(data))
            f.write(head)
            f.write(struct.pack('<I', masked_crc(head)))
            f.write(data)
            f.write(struct.pack('<I', masked_crc(data)))


# ------------------------------------------------------------------ protobuf wire format (the subset Example uses)
def _varint(buf, pos):
    out = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7f) << shift
        if not b & 0x80:
            return out, pos
        shift += 7


def _fields(buf):
    """Yield (field number, wire type, value) of one message; length-delimited values as memoryview slices."""
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        num, wt = key >> 3, key & 7
        if wt == 0:
            val, pos = _varint(buf, pos)
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            val = buf[pos:pos + ln]
            pos += ln
        elif wt == 5:
            val = buf[pos:pos + 4]
            pos += 4
        elif wt == 1:
            val = buf[pos:pos + 8]
            pos += 8
        else:
            raise ValueError('unsupported protobuf wire type %d' % wt)
        yield num, wt, val


_VIEW_FROM = 4096       # bytes values at least this long stay views when the caller asked for views (the raw images)


def _feature(buf, views=False):
    for num, _wt, val in _fields(buf):
        if num == 1:        # BytesList
            return [(v if views and len(v) >= _VIEW_FROM else bytes(v)) for n, _w, v in _fields(val) if n == 1]
        if num == 2:        # FloatList: packed or repeated fixed32
            out = []
            for n, w, v in _fields(val):
                if n == 1:
                    out.extend(np.frombuffer(bytes(v), dtype='<f4').tolist())
            return out
        if num == 3:        # Int64List: packed or repeated varint
            out = []
            for n, w, v in _fields(val):
                if n != 1:
                    continue
                if w == 0:
                    out.append(v)
                else:
                    p, m = 0, memoryview(bytes(v))
                    while p < len(m):
                        x, p = _varint(m, p)
                        out.append(x)
            return [x - (1 << 64) if x >= (1 << 63) else x for x in out]
    return []


def parse_example(data, views=False):
    """Serialized tf.train.Example -> {feature name: list of bytes / floats / ints}.  views: long bytes values (the raw
    images) come back as memoryview slices of ``data`` instead of copies."""
    out = {}
    for num, _wt, feats in _fields(memoryview(data)):
        if num != 1:
            continue
        for n2, _w2, entry in _fields(feats):          # map<string, Feature> entries
            if n2 != 1:
                continue
            key, value = None, []
            for n3, _w3, v in _fields(entry):
                if n3 == 1:
                    key = bytes(v).decode('utf-8')
                elif n3 == 2:
                    value = _feature(v, views)
            out[key] = value
    return out


# ------------------------------------------------------------------ writer side (tests and dataset conversion)
def _enc_varint(x):
    x &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = x & 0x7f
        x >>= 7
        if x:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _ld(num, payload):
    return _enc_varint((num << 3) | 2) + _enc_varint(len(payload)) + payload


def make_example(features):
    """{name: bytes | int | float | list of those} -> serialized tf.train.Example."""
    body = b''
    for name in sorted(features):
        v = features[name]
        vals = v if isinstance(v, (list, tuple)) else [v]
        if isinstance(vals[0], (bytes, bytearray)):
            feat = _ld(1, b''.join(_ld(1, bytes(x)) for x in vals))
        elif isinstance(vals[0], float):
            feat = _ld(2, _ld(1, np.asarray(vals, dtype='<f4').tobytes()))
        else:
            feat = _ld(3, _ld(1, b''.join(_enc_varint(int(x)) for x in vals)))
        body += _ld(1, _ld(1, name.encode('utf-8')) + _ld(2, feat))
    return _ld(1, body)
