"""Independent writer of TensorFlow V2 checkpoints (tensor bundles) for the tests: the same format knowledge as
make_format_fixtures.py (LevelDB table_format.md + tensor_bundle.proto), its OWN varint / protobuf / block / footer code and its
own CRC-32C -- nothing from the package under test, no TensorFlow -- as a function, so that a test can write a whole model
(~100 MB: too large to commit as a fixture) the way tf.train.Saver lays it out: header entry "", one BundleEntryProto per
variable sorted by name, 16-entry restart intervals, ~4 KiB data blocks, one index block, the 48-byte footer.

The CRC of a 100 MB data shard in pure Python would take minutes: the tensor checksums come from a 15-line C function compiled
with gcc on first use (bytewise table form of the same reflected polynomial 0x82F63B78); the pure-Python bitwise form is kept
for short inputs and is what the C form is checked against."""
import ctypes
import os
import struct
import subprocess
import tempfile

import numpy as np

DT_FLOAT, DT_INT32, DT_INT64 = 1, 3, 9
_C_SRC = r'''
#include <stddef.h>
#include <stdint.h>
uint32_t crc32c_bytes(const uint8_t* p, size_t n) {
    static uint32_t T[256];
    static int ready = 0;
    if (!ready) {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
            T[i] = c;
        }
        ready = 1;
    }
    uint32_t c = 0xFFFFFFFFu;
    while (n--) c = T[(c ^ *p++) & 0xffu] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}
'''
_clib = None


def crc32c_bitwise(data):
    crc = 0xFFFFFFFF
    for b in data:
        crc ^= b
        for _ in range(8):
            crc = (crc >> 1) ^ (0x82F63B78 if crc & 1 else 0)
    return crc ^ 0xFFFFFFFF


def crc32c(data):
    global _clib
    if len(data) < 4096:
        return crc32c_bitwise(data)
    if _clib is None:
        d = tempfile.mkdtemp(prefix='crc32c_ref_')
        src, so = os.path.join(d, 'c.c'), os.path.join(d, 'c.so')
        open(src, 'w').write(_C_SRC)
        subprocess.check_call(['gcc', '-O2', '-shared', '-fPIC', src, '-o', so])
        _clib = ctypes.CDLL(so)
        _clib.crc32c_bytes.restype = ctypes.c_uint32
        _clib.crc32c_bytes.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
        probe = bytes(range(256)) * 20
        assert _clib.crc32c_bytes(probe, len(probe)) == crc32c_bitwise(probe)
    return int(_clib.crc32c_bytes(data, len(data)))


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


def block(kvs, restart_interval=16):
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
    if not restarts:
        restarts = [0]
    return out + b''.join(struct.pack('<I', r) for r in restarts) + struct.pack('<I', len(restarts))


def with_trailer(body):
    return body + b'\x00' + struct.pack('<I', masked(crc32c(body + b'\x00')))   # type 0 = no compression


def write_bundle(prefix, tensors, block_bytes=4096):
    """tensors: {variable name: ndarray (float32 / int32 / int64)} -> prefix.index + prefix.data-00000-of-00001."""
    entries = []
    offset = 0
    with open(prefix + '.data-00000-of-00001', 'wb') as data:
        for name in sorted(tensors, key=lambda s: s.encode()):
            arr = np.asarray(tensors[name])      # (ascontiguousarray would turn a scalar into shape (1,))
            dt = {np.dtype('float32'): DT_FLOAT, np.dtype('int32'): DT_INT32, np.dtype('int64'): DT_INT64}[arr.dtype]
            raw = arr.tobytes(order='C')
            shape = b''.join(field(2, 2, field(1, 0, int(d))) for d in arr.shape)      # TensorShapeProto.dim(2) { size(1) }
            e = field(1, 0, dt) + field(2, 2, shape)
            if offset:
                e += field(4, 0, offset)                                                # proto3: a zero offset is omitted
            e += field(5, 0, len(raw)) + field(6, 5, masked(crc32c(raw)))               # size, crc32c (fixed32, masked)
            entries.append((name.encode(), e))
            data.write(raw)
            offset += len(raw)
    header = field(1, 0, 1) + field(3, 2, field(1, 0, 1))                               # num_shards = 1, version { producer = 1 }
    entries = [(b'', header)] + entries
    table, index_entries, chunk, size = b'', [], [], 0
    for kv in entries + [None]:
        if kv is not None:
            chunk.append(kv)
            size += len(kv[0]) + len(kv[1]) + 3
        if chunk and (kv is None or size >= block_bytes):
            body = block(chunk)
            index_entries.append((chunk[-1][0], varint(len(table)) + varint(len(body))))
            table += with_trailer(body)
            chunk, size = [], 0
    meta_body = block([], 1)
    meta_handle = varint(len(table)) + varint(len(meta_body))
    table += with_trailer(meta_body)
    index_body = block(index_entries, 1)
    index_handle = varint(len(table)) + varint(len(index_body))
    table += with_trailer(index_body)
    footer = meta_handle + index_handle
    table += footer + b'\x00' * (40 - len(footer)) + struct.pack('<Q', 0xdb4775248b80fb57)
    with open(prefix + '.index', 'wb') as f:
        f.write(table)
    return len(table), offset
