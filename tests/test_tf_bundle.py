"""TF-V2 checkpoint reader / writer and the averaging tool.  No TensorFlow exists in the
build image and the reference ships no checkpoint, so these tests pin the format against hand-built bytes
(crc32c known answers, a LevelDB block written by hand) and the reader against the writer: PARITY UNPINNED
against a TF-written file."""
import os
import struct

import numpy as np
import pytest

from asr_amd import avg_checkpoints, tf_bundle, weights
from tests.helpers import product_config, tiny_config


def test_crc32c_known_answers():
    # RFC 3720 B.4 test vectors
    assert tf_bundle.crc32c(b'\x00' * 32) == 0x8a9136aa
    assert tf_bundle.crc32c(b'\xff' * 32) == 0x62a8ab43
    assert tf_bundle.crc32c(bytes(range(32))) == 0x46dd794e
    assert tf_bundle.crc32c(b'123456789') == 0xe3069283
    assert tf_bundle.crc32c(b'6789', tf_bundle.crc32c(b'12345')) == 0xe3069283
    big = bytes(range(256)) * 100                                   # > 4096 bytes: the library's host routine
    c = 0
    for i in range(0, len(big), 1000):
        c = tf_bundle.crc32c(big[i:i + 1000], c)
    assert tf_bundle.crc32c(big) == c
    # LevelDB's mask
    assert tf_bundle.mask_crc(0) == 0xa282ead8


def test_block_prefix_compression_hand_built():
    # entries ("apple","1"), ("apply","22"), ("b","") with a restart at 0: shared/non_shared/value_len bytes
    blk = bytes([0, 5, 1]) + b'apple' + b'1' + bytes([4, 1, 2]) + b'y' + b'22' + bytes([0, 1, 0]) + b'b' + \
        struct.pack('<II', 0, 1)
    assert tf_bundle._block_entries(blk) == [(b'apple', b'1'), (b'apply', b'22'), (b'b', b'')]
    assert tf_bundle._block_entries(tf_bundle._build_block([(b'apple', b'1'), (b'apply', b'22'), (b'b', b'')])) == \
        [(b'apple', b'1'), (b'apply', b'22'), (b'b', b'')]


def test_entry_proto_hand_built():
    # dtype DT_FLOAT(1), shape [3, 4], offset 48, size 48, crc fixed32
    msg = bytes([0x08, 1, 0x12, 8, 0x12, 2, 0x08, 3, 0x12, 2, 0x08, 4, 0x20, 48, 0x28, 48, 0x35]) + struct.pack('<I', 0xdeadbeef)
    e = tf_bundle._parse_entry(msg)
    assert (e['dtype'], e['shape'], e['offset'], e['size'], e['crc32c'], e['shard_id']) == (1, (3, 4), 48, 48, 0xdeadbeef, 0)
    assert tf_bundle._entry_proto(1, (3, 4), 48, 48, 0xdeadbeef) == msg


def test_round_trip_with_optimizer_slots(tmp_path):
    cfg = product_config(tiny_config())
    W = weights.random_init(cfg, seed=3)
    ck = dict(W)
    ck['global_step'] = np.array(1234, np.int64)
    ck['beta1_power'] = np.array(0.5, np.float32)
    ck['encoder/src_change/kernel/Adam'] = np.ones_like(W['encoder/src_change/kernel'])
    prefix = str(tmp_path / 'model_step_1234')
    tf_bundle.write_bundle(prefix, ck)
    tf_bundle.write_checkpoint_state(str(tmp_path), 'model_step_1234')
    assert os.path.exists(prefix + '.index') and os.path.exists(prefix + '.data-00000-of-00001')
    back = tf_bundle.read_bundle(prefix, verify_crc=True)
    assert set(back) == set(ck)
    for k in ck:
        assert back[k].dtype == ck[k].dtype and back[k].shape == ck[k].shape and np.array_equal(back[k], ck[k])
    assert dict(tf_bundle.list_variables(prefix))['encoder/src_change/kernel'] == [128, 320]
    # the drop-in path: latest_checkpoint + load_checkpoint drop the optimizer state and keep float32 graph variables
    p = weights.latest_checkpoint(str(tmp_path))
    assert p == prefix
    got = weights.load_checkpoint(p)
    assert set(got) == set(W)
    assert all(np.array_equal(got[k], W[k]) for k in W)


def test_corruption_is_detected(tmp_path):
    prefix = str(tmp_path / 'c')
    tf_bundle.write_bundle(prefix, {'a': np.arange(12, dtype=np.float32).reshape(3, 4)})
    raw = bytearray(open(prefix + '.data-00000-of-00001', 'rb').read())
    raw[5] ^= 1
    open(prefix + '.data-00000-of-00001', 'wb').write(bytes(raw))
    with pytest.raises(ValueError):
        tf_bundle.read_bundle(prefix, verify_crc=True)
    idx = bytearray(open(prefix + '.index', 'rb').read())
    idx[3] ^= 1
    open(prefix + '.index', 'wb').write(bytes(idx))
    with pytest.raises(ValueError):
        tf_bundle.read_index(prefix + '.index')
    open(prefix + '.index', 'wb').write(b'not a table' * 10)
    with pytest.raises(ValueError):
        tf_bundle.read_index(prefix + '.index')


def test_avg_checkpoints_matches_reference_arithmetic(tmp_path):
    cfg = product_config(tiny_config())
    cks = []
    for i in range(3):
        W = weights.random_init(cfg, seed=10 + i)
        W = {k: (v + np.float32(0.01 * i)) for k, v in W.items()}
        W['global_step'] = np.array(100 * i, np.int64)
        W['beta2_power'] = np.array(0.9 ** i, np.float32)
        name = 'model_step_%d' % (100 * i)
        if i == 1:
            np.savez(str(tmp_path / name) + '.npz', **W)            # mixed input formats
        else:
            tf_bundle.write_bundle(str(tmp_path / name), W)
        cks.append(W)
    out = avg_checkpoints.main(['--checkpoints', 'model_step_0, model_step_100,model_step_200,missing_one',
                                '--prefix', str(tmp_path) + '/', '--output_path', str(tmp_path / 'averaged.ckpt')])
    assert out.endswith('averaged.ckpt-0')
    assert weights.latest_checkpoint(str(tmp_path)) == out           # avg_checkpoints.py:107 + evaluate.py:36
    got = tf_bundle.read_bundle(out)
    assert got['global_step'].dtype == np.int64 and int(got['global_step']) == 0
    for k in cks[0]:
        if k.startswith('global_step'):
            continue
        want = (sum(np.zeros(c[k].shape) + c[k] for c in cks) / 3).astype(np.float32)   # float64 sum, then cast
        assert np.array_equal(got[k], want), k
    loaded = weights.load_checkpoint(out)                             # beta2_power / global_step dropped
    assert 'beta2_power' not in loaded and 'global_step' not in loaded
    with np.load(out + '.npz') as z:
        assert np.array_equal(z['encoder/src_change/kernel'], got['encoder/src_change/kernel'])
    with pytest.raises(ValueError):
        avg_checkpoints.main(['--checkpoints', 'nope', '--prefix', str(tmp_path) + '/'])
