"""Hand-assembled byte vectors for the file formats either side of the path:
a whole TF-V2 checkpoint `.index` + data shard written byte by byte in this file (independent of
asr_amd.tf_bundle.write_bundle), an 8-bit `BCM` Kaldi matrix, and the reference's own YAML files through
load_config.  CPU only."""
import glob
import os
import struct

import numpy as np
import pytest

from asr_amd import ark, tf_bundle, weights
from asr_amd.evaluate import load_config
from asr_amd.model import asr_config_from
from asr_amd import _lib

REF_CFG_DIR = '/root/reference/transformer'


# ---- an independent crc32c (bitwise, no table) + LevelDB mask, checked against RFC 3720 B.4 -----------------
def _crc32c_bitwise(data):
    c = 0xFFFFFFFF
    for b in data:
        c ^= b
        for _ in range(8):
            c = (c >> 1) ^ (0x82F63B78 if c & 1 else 0)
    return c ^ 0xFFFFFFFF


def _masked(c):
    return (((c >> 15) | (c << 17)) + 0xa282ead8) & 0xFFFFFFFF


def test_bitwise_crc_known_answers():
    assert _crc32c_bitwise(b'123456789') == 0xe3069283
    assert _crc32c_bitwise(bytes(range(32))) == 0x46dd794e


def _block_with_trailer(contents):
    """LevelDB block on disk: contents + 1 type byte (0 = uncompressed) + masked crc32c(contents + type)."""
    return contents + b'\x00' + struct.pack('<I', _masked(_crc32c_bitwise(contents + b'\x00')))


def _hand_built_checkpoint(tmp_path):
    """Three variables, written as TensorFlow's BundleWriter lays them out:
         enc/bias    float32 [3]    offset 0   size 12
         enc/kernel  float32 [2,3]  offset 12  size 24
         global_step int64   []     offset 36  size 8
    """
    bias = np.array([0.5, -1.25, 3.0], '<f4')
    kernel = np.array([[1, 2, 3], [4, 5, 6.5]], '<f4')
    step = np.array(4321, '<i8')
    data = bias.tobytes() + kernel.tobytes() + step.tobytes()
    assert len(data) == 44

    def entry(dtype, dims, offset, size, raw):
        # BundleEntryProto: 1 dtype, 2 shape{2 dim{1 size}}, 4 offset (omitted when 0), 5 size, 6 crc32c (fixed32, masked)
        shape = b''.join(bytes([0x12, 0x02, 0x08, d]) for d in dims)
        msg = bytes([0x08, dtype, 0x12, len(shape)]) + shape
        if offset:
            msg += bytes([0x20, offset])
        msg += bytes([0x28, size, 0x35]) + struct.pack('<I', _masked(_crc32c_bitwise(raw)))
        return msg

    e_bias = entry(1, [3], 0, 12, bias.tobytes())                # DT_FLOAT = 1
    e_kernel = entry(1, [2, 3], 12, 24, kernel.tobytes())
    e_step = entry(9, [], 36, 8, step.tobytes())                 # DT_INT64 = 9
    header = bytes([0x08, 0x01, 0x1a, 0x02, 0x08, 0x01])         # BundleHeaderProto: num_shards 1, version{producer 1}
    # data block: entries (shared, non_shared, value_len, key delta, value), prefix-compressed against the previous
    # key, one restart point at offset 0, then the restart array and its length
    blk = bytes([0, 0, len(header)]) + header
    blk += bytes([0, 8, len(e_bias)]) + b'enc/bias' + e_bias
    blk += bytes([4, 6, len(e_kernel)]) + b'kernel' + e_kernel               # shares "enc/" with "enc/bias"
    blk += bytes([0, 11, len(e_step)]) + b'global_step' + e_step
    blk += struct.pack('<II', 0, 1)
    out = bytearray()
    data_off, data_size = 0, len(blk)
    out += _block_with_trailer(blk)
    meta = struct.pack('<II', 0, 1)                              # empty metaindex block
    meta_off = len(out)
    out += _block_with_trailer(meta)
    # index block: one entry, key = a separator >= the data block's last key ("h" = shortest successor of
    # "global_step"), value = the data block's handle (varint offset, varint size); restart interval 1
    assert data_size < 128 * 128
    handle = bytes([data_off]) + (bytes([data_size]) if data_size < 128 else bytes([(data_size & 0x7f) | 0x80, data_size >> 7]))
    idx = bytes([0, 1, len(handle)]) + b'h' + handle + struct.pack('<II', 0, 1)
    idx_off = len(out)
    out += _block_with_trailer(idx)

    def varint(v):
        b = bytearray()
        while v >= 128:
            b.append((v & 0x7f) | 0x80)
            v >>= 7
        b.append(v)
        return bytes(b)
    footer = varint(meta_off) + varint(len(meta)) + varint(idx_off) + varint(len(idx))
    out += footer + b'\x00' * (40 - len(footer)) + bytes.fromhex('57fb808b247547db')   # table magic, little endian
    prefix = str(tmp_path / 'model_step_4321')
    with open(prefix + '.index', 'wb') as f:
        f.write(bytes(out))
    with open(prefix + '.data-00000-of-00001', 'wb') as f:
        f.write(data)
    with open(str(tmp_path / 'checkpoint'), 'w') as f:
        f.write('model_checkpoint_path: "model_step_4321"\nall_model_checkpoint_paths: "model_step_4321"\n')
    return prefix, dict(bias=bias, kernel=kernel, step=step)


def test_whole_index_file_assembled_by_hand(tmp_path):
    prefix, want = _hand_built_checkpoint(tmp_path)
    header, entries = tf_bundle.read_index(prefix + '.index')
    assert header['num_shards'] == 1 and sorted(entries) == ['enc/bias', 'enc/kernel', 'global_step']
    assert entries['enc/kernel']['shape'] == (2, 3) and entries['enc/kernel']['offset'] == 12 and entries['enc/kernel']['size'] == 24
    assert entries['global_step']['shape'] == () and entries['global_step']['dtype'] == 9
    got = tf_bundle.read_bundle(prefix, verify_crc=True)
    assert got['enc/bias'].dtype == np.float32 and np.array_equal(got['enc/bias'], want['bias'])
    assert np.array_equal(got['enc/kernel'], want['kernel'])
    assert got['global_step'].dtype == np.int64 and int(got['global_step']) == 4321
    assert tf_bundle.list_variables(prefix) == [('enc/bias', [3]), ('enc/kernel', [2, 3]), ('global_step', [])]
    # the drop-in path (evaluate.py:36): latest_checkpoint -> load_checkpoint keeps the float variables only
    assert weights.latest_checkpoint(str(tmp_path)) == prefix
    loaded = weights.load_checkpoint(prefix)
    assert sorted(loaded) == ['enc/bias', 'enc/kernel']
    # and the writer emits the very same index bytes for the same variables (writer pinned by the hand-built file)
    p2 = str(tmp_path / 'again')
    tf_bundle.write_bundle(p2, {'enc/bias': want['bias'], 'enc/kernel': want['kernel'], 'global_step': want['step']})
    assert open(p2 + '.data-00000-of-00001', 'rb').read() == open(prefix + '.data-00000-of-00001', 'rb').read()
    h2, e2 = tf_bundle.read_index(p2 + '.index')
    assert e2 == entries and h2 == header


def test_hand_built_index_detects_a_flipped_bit(tmp_path):
    prefix, _ = _hand_built_checkpoint(tmp_path)
    raw = bytearray(open(prefix + '.index', 'rb').read())
    raw[10] ^= 0x04
    open(prefix + '.index', 'wb').write(bytes(raw))
    with pytest.raises(ValueError):
        tf_bundle.read_index(prefix + '.index')


def test_ark_bcm_8bit_hand_built(tmp_path):
    """Kaldi 8-bit compressed matrix (third_party/feat_convert/io/ark.py:79-108): global header (min, range, rows,
    cols), per-column uint16 percentiles (p0, p25, p75, p100), then the bytes COLUMN by column."""
    rows, cols = 4, 2
    gmin, grange = -2.0, 8.0
    pct = [(0, 16384, 49152, 65535), (6553, 13107, 26214, 52428)]
    col_bytes = [[0, 64, 192, 255], [10, 100, 200, 65]]
    p = str(tmp_path / 'c.ark')
    with open(p, 'wb') as f:
        f.write(b'utt_c ')
        off = f.tell()
        f.write(b'\0BCM ' + struct.pack('<ffii', gmin, grange, rows, cols))
        for q in pct:
            f.write(struct.pack('<HHHH', *q))
        for c in col_bytes:
            f.write(bytes(c))
    with open(str(tmp_path / 'c.scp'), 'w') as f:
        f.write('utt_c %s:%d\n' % (p, off))
    got = ark.ArkReader(str(tmp_path / 'c.scp')).read_utt('utt_c')
    assert got.shape == (rows, cols) and got.dtype == np.float32
    # the reference's loop, written out (ark.py:91-108)
    want = np.zeros((rows, cols), np.float32)
    for i in range(cols):
        p0, p25, p75, p100 = (float(gmin + grange * x / 65535.0) for x in pct[i])
        d1, d2, d3 = (p25 - p0) / 64.0, (p75 - p25) / 128.0, (p100 - p75) / 63.0
        for j in range(rows):
            c = col_bytes[i][j]
            want[j][i] = p0 + d1 * c if c <= 64 else (p25 + d2 * (c - 64) if c <= 192 else p75 + d3 * (c - 192))
    assert np.array_equal(got, want)
    # exact landmarks: byte 0 -> p0, 64 -> p25, 192 -> p75, 255 -> p100 of column 0
    assert got[0, 0] == np.float32(-2.0) and got[3, 0] == np.float32(6.0)
    assert abs(float(got[1, 0]) - (-2.0 + 8.0 * 16384 / 65535.0)) < 1e-6


@pytest.mark.skipif(not os.path.isdir(REF_CFG_DIR), reason='reference tree not present (GPU box)')
def test_reference_yaml_templates_load_unchanged():
    """Every transformer/config_template_*.yaml of the reference goes through load_config unchanged and derives the
    asr_config the library is created with (duplicate `set1` key of ...seame_sp.yaml: YAML last-wins; beam_size 13)."""
    files = sorted(glob.glob(os.path.join(REF_CFG_DIR, 'config_template_*.yaml')))
    assert len(files) == 7
    vocab = {}
    for f in files:
        cfg = load_config(f)
        c, prec = asr_config_from(cfg, device=0)
        assert (c.input_dim, c.hidden_units, c.num_heads, c.num_blocks) == (320, 1024, 16, 6), f
        assert c.ff_activation == _lib.FF_ACTIVATIONS['glu'] and c.scale_embedding == 1 and c.tie_embedding_and_softmax == 1
        assert c.beam_size == 13 and c.max_target_length == 200 and abs(c.lp_alpha - 0.6) < 1e-7
        assert prec == os.environ.get('ASR_AMD_PRECISION', 'f32_f16x3') and c.sub_batches == 0
        assert cfg.model == 'Transformer' and cfg.no_such_key is None
        sets = [k for k in cfg.test if k.startswith('set')]
        assert sets and all(cfg.test[s]['src_path'] and cfg.test[s]['output_path'] for s in sets)
        vocab[os.path.basename(f)] = c.dst_vocab_size
        assert len(weights.variable_shapes(cfg)) == 4 + 6 * 14 + 1 + 6 * 22
    assert sorted(vocab.values()) == [122, 1388, 3205, 3205, 3900, 11039, 28444]
    seame = load_config(os.path.join(REF_CFG_DIR, 'config_template_char_left3_big_dim80_seame.yaml'))
    assert seame.test.tokens_per_batch == 7000 and seame.test.num_gpus == 8
    sp = load_config(os.path.join(REF_CFG_DIR, 'config_template_char_left3_big_dim80_seame_sp.yaml'))
    assert [k for k in sp.test if k.startswith('set')] == ['set1']
