"""TensorFlow "V2" checkpoint (tensor bundle) reader / writer without TensorFlow.

What `model.saver.restore(sess, tf.train.latest_checkpoint(model_dir))` (evaluate.py:36) reads and
`tf.train.Saver(...).save` (train.py:66-67, avg_checkpoints.py:98-107) writes:

    <prefix>.index                 an SSTable (LevelDB table format, uncompressed blocks): key "" ->
                                   BundleHeaderProto, key <variable name> -> BundleEntryProto
    <prefix>.data-SSSSS-of-NNNNN   the raw tensor bytes, little endian, at (shard_id, offset, size)

TensorFlow is not vendored in the reference and not installed in the build image, so the format below is
restated from its published on-disk format (tensorflow/core/util/tensor_bundle, lib/io/table format ==
LevelDB's): PARITY UNPINNED against a TF-written file; the reader and the writer here are tested against
each other and against hand-built byte strings (tests/test_tf_bundle.py).  Only what the reference's
checkpoints contain is supported: dense, unpartitioned variables (no `slices`), no snappy blocks.
"""
from __future__ import annotations

import os
import struct

import numpy as np

TABLE_MAGIC = 0xdb4775248b80fb57
# tensorflow DataType enum -> numpy
DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8, 9: np.int64, 10: np.bool_,
          17: np.uint16, 19: np.float16, 22: np.uint32, 23: np.uint64}
DTYPE_CODES = {np.dtype(v): k for k, v in DTYPES.items()}


# ---- crc32c (Castagnoli), masked as LevelDB / TF store it ----------------------------------------
def _make_table():
    t = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        t.append(c)
    return np.array(t, dtype=np.uint32)


_CRC_TABLE = _make_table()


def crc32c(data, crc=0):
    data = bytes(data)
    if len(data) > 4096:                    # large tensors: the library's host routine (same polynomial, no GPU)
        try:
            from ._lib import lib
            return int(lib.asr_crc32c(data, len(data), crc))
        except (ImportError, OSError, AttributeError):
            pass
    c = (~crc) & 0xFFFFFFFF
    t = _CRC_TABLE
    for b in bytes(data):
        c = int(t[(c ^ b) & 0xFF]) ^ (c >> 8)
    return (~c) & 0xFFFFFFFF


def mask_crc(c):
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xa282ead8) & 0xFFFFFFFF


# ---- varints / protobuf wire format ----------------------------------------------------------------
def _read_varint(buf, pos):
    shift = result = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7
        if shift > 70:
            raise ValueError('malformed varint')


def _write_varint(v):
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _parse_fields(buf):
    """protobuf message -> list of (field number, wire type, value); value is int or bytes."""
    pos, out = 0, []
    while pos < len(buf):
        tag, pos = _read_varint(buf, pos)
        field, wt = tag >> 3, tag & 7
        if wt == 0:
            v, pos = _read_varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from('<Q', buf, pos)[0]
            pos += 8
        elif wt == 2:
            n, pos = _read_varint(buf, pos)
            v = bytes(buf[pos:pos + n])
            pos += n
        elif wt == 5:
            v = struct.unpack_from('<I', buf, pos)[0]
            pos += 4
        else:
            raise ValueError('unsupported protobuf wire type %d' % wt)
        out.append((field, wt, v))
    return out


def _signed64(v):
    return v - (1 << 64) if v >= (1 << 63) else v


def _parse_shape(buf):
    dims = []
    for f, wt, v in _parse_fields(buf):
        if f == 2 and wt == 2:                                  # repeated Dim dim = 2
            size = 0
            for f2, wt2, v2 in _parse_fields(v):
                if f2 == 1 and wt2 == 0:
                    size = _signed64(v2)
            dims.append(size)
        elif f == 3 and wt == 0 and v:
            raise ValueError('unknown-rank tensor in checkpoint')
    return tuple(dims)


def _parse_entry(buf):
    e = {'dtype': 0, 'shape': (), 'shard_id': 0, 'offset': 0, 'size': 0, 'crc32c': None, 'sliced': False}
    for f, wt, v in _parse_fields(buf):
        if f == 1:
            e['dtype'] = v
        elif f == 2:
            e['shape'] = _parse_shape(v)
        elif f == 3:
            e['shard_id'] = v
        elif f == 4:
            e['offset'] = v
        elif f == 5:
            e['size'] = v
        elif f == 6:
            e['crc32c'] = v
        elif f == 7:
            e['sliced'] = True
    return e


# ---- LevelDB table ----------------------------------------------------------------------------------
def _read_block(data, offset, size, verify=True):
    contents = data[offset:offset + size]
    trailer = data[offset + size:offset + size + 5]
    if len(contents) != size or len(trailer) != 5:
        raise ValueError('truncated table block')
    if trailer[0] != 0:
        raise ValueError('compressed table block (type %d): only uncompressed index files are supported' % trailer[0])
    if verify:
        want = struct.unpack('<I', trailer[1:5])[0]
        if mask_crc(crc32c(trailer[:1], crc32c(contents))) != want:
            raise ValueError('table block checksum mismatch')
    return contents


def _block_entries(block):
    if len(block) < 4:
        raise ValueError('bad table block')
    n_restarts = struct.unpack_from('<I', block, len(block) - 4)[0]
    end = len(block) - 4 - 4 * n_restarts
    pos, key, out = 0, b'', []
    while pos < end:
        shared, pos = _read_varint(block, pos)
        non_shared, pos = _read_varint(block, pos)
        vlen, pos = _read_varint(block, pos)
        key = key[:shared] + bytes(block[pos:pos + non_shared])
        pos += non_shared
        out.append((key, bytes(block[pos:pos + vlen])))
        pos += vlen
    return out


def read_index(path, verify=True):
    """<prefix>.index -> (header dict, {name: entry dict})."""
    with open(path, 'rb') as f:
        data = f.read()
    if len(data) < 48 or struct.unpack('<Q', data[-8:])[0] != TABLE_MAGIC:
        raise ValueError('%s is not a TensorFlow V2 checkpoint index (bad table magic)' % path)
    footer = data[-48:]
    pos = 0
    _, pos = _read_varint(footer, pos)                          # metaindex handle
    _, pos = _read_varint(footer, pos)
    ioff, pos = _read_varint(footer, pos)                       # index handle
    isize, pos = _read_varint(footer, pos)
    header, entries = {'num_shards': 1, 'endianness': 0}, {}
    for _, handle in _block_entries(_read_block(data, ioff, isize, verify)):
        boff, p = _read_varint(handle, 0)
        bsize, p = _read_varint(handle, p)
        for key, value in _block_entries(_read_block(data, boff, bsize, verify)):
            if key == b'':
                for f, wt, v in _parse_fields(value):
                    if f == 1:
                        header['num_shards'] = v
                    elif f == 2:
                        header['endianness'] = v
            else:
                entries[key.decode('utf-8')] = _parse_entry(value)
    if header['endianness'] != 0:
        raise ValueError('big-endian checkpoint')
    return header, entries


def list_variables(prefix):
    """tf.contrib.framework.list_variables: [(name, shape)] sorted by name."""
    _, entries = read_index(prefix + '.index')
    return [(n, list(e['shape'])) for n, e in sorted(entries.items())]


def read_bundle(prefix, names=None, verify_crc=False):
    """Checkpoint prefix -> {variable name: numpy array}.  `names` restricts what is read.
    verify_crc checks every tensor's crc32c (pure Python: slow on large checkpoints)."""
    header, entries = read_index(prefix + '.index')
    out, files = {}, {}
    try:
        for name in sorted(entries):
            if names is not None and name not in names:
                continue
            e = entries[name]
            if e['sliced']:
                raise ValueError('partitioned variable %s is not supported' % name)
            if e['dtype'] not in DTYPES:
                continue                                        # strings / resources: not part of this model
            dt = np.dtype(DTYPES[e['dtype']])
            count = int(np.prod(e['shape'], dtype=np.int64)) if e['shape'] else 1
            if count * dt.itemsize != e['size']:
                raise ValueError('%s: size %d does not match shape %s' % (name, e['size'], e['shape']))
            sid = e['shard_id']
            if sid not in files:
                files[sid] = open('%s.data-%05d-of-%05d' % (prefix, sid, header['num_shards']), 'rb')
            fh = files[sid]
            fh.seek(e['offset'])
            raw = fh.read(e['size'])
            if len(raw) != e['size']:
                raise ValueError('%s: data shard truncated' % name)
            if verify_crc and e['crc32c'] is not None and mask_crc(crc32c(raw)) != e['crc32c']:
                raise ValueError('%s: tensor checksum mismatch' % name)
            out[name] = np.frombuffer(raw, dtype=dt).reshape(e['shape']).copy()
    finally:
        for fh in files.values():
            fh.close()
    return out


# ---- writer (single shard, single data block) ---------------------------------------------------------
def _field(num, wt, payload):
    return _write_varint((num << 3) | wt) + payload


def _entry_proto(dtype_code, shape, offset, size, crc):
    shape_msg = b''.join(_field(2, 2, _write_varint(len(d)) + d)
                         for d in (_field(1, 0, _write_varint(s)) for s in shape))
    msg = _field(1, 0, _write_varint(dtype_code))
    msg += _field(2, 2, _write_varint(len(shape_msg)) + shape_msg)
    if offset:
        msg += _field(4, 0, _write_varint(offset))
    msg += _field(5, 0, _write_varint(size))
    msg += _field(6, 5, struct.pack('<I', crc))
    return msg


def _build_block(items, restart_interval=16):
    out, restarts, prev = bytearray(), [], b''
    for i, (key, value) in enumerate(items):
        shared = 0
        if i % restart_interval == 0:
            restarts.append(len(out))
        else:
            while shared < min(len(prev), len(key)) and prev[shared] == key[shared]:
                shared += 1
        out += _write_varint(shared) + _write_varint(len(key) - shared) + _write_varint(len(value))
        out += key[shared:] + value
        prev = key
    if not restarts:
        restarts = [0]
    for r in restarts:
        out += struct.pack('<I', r)
    out += struct.pack('<I', len(restarts))
    return bytes(out)


def _emit_block(buf, block):
    off = len(buf)
    buf += block + b'\x00' + struct.pack('<I', mask_crc(crc32c(b'\x00', crc32c(block))))
    return _write_varint(off) + _write_varint(len(block))


def write_bundle(prefix, variables):
    """{name: array} -> <prefix>.index + <prefix>.data-00000-of-00001 (what Saver.save writes)."""
    d = os.path.dirname(prefix)
    if d:
        os.makedirs(d, exist_ok=True)
    items = [(b'', _field(1, 0, _write_varint(1)) + _field(3, 2, _write_varint(2) + _field(1, 0, _write_varint(1))))]
    offset = 0
    with open(prefix + '.data-00000-of-00001', 'wb') as f:
        for name in sorted(variables, key=lambda s: s.encode('utf-8')):
            a = np.asarray(variables[name])                     # (ascontiguousarray would turn a scalar into shape (1,))
            if a.dtype not in DTYPE_CODES:
                raise ValueError('%s: dtype %s cannot be stored' % (name, a.dtype))
            raw = a.astype(a.dtype.newbyteorder('<'), copy=False).tobytes()
            f.write(raw)
            items.append((name.encode('utf-8'), _entry_proto(DTYPE_CODES[a.dtype], a.shape, offset, len(raw),
                                                             mask_crc(crc32c(raw)))))
            offset += len(raw)
    buf = bytearray()
    data_handle = _emit_block(buf, _build_block(items))
    meta_handle = _emit_block(buf, _build_block([]))
    index_handle = _emit_block(buf, _build_block([(items[-1][0] + b'\x00', data_handle)], restart_interval=1))
    footer = meta_handle + index_handle
    buf += footer + b'\x00' * (40 - len(footer)) + struct.pack('<Q', TABLE_MAGIC)
    with open(prefix + '.index', 'wb') as f:
        f.write(bytes(buf))


def write_checkpoint_state(model_dir, basename):
    """The `checkpoint` text file tf.train.latest_checkpoint reads."""
    with open(os.path.join(model_dir, 'checkpoint'), 'w') as f:
        f.write('model_checkpoint_path: "%s"\nall_model_checkpoint_paths: "%s"\n' % (basename, basename))
