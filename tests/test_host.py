"""CPU tests of the host-side mirror of evaluate.py's harness (no GPU, no compute calls)."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import yaml

from asr_amd import ark, weights
from asr_amd.utils import AttrDict, DataReader, expand_feed_dict, shard_bounds
from oracle import asr_oracle as O
from tests.helpers import product_config, tiny_config

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_attrdict_semantics():
    d = AttrDict({'a': 1, 'test': {'beam_size': 13, 'set1': {'src_path': 'x'}}})
    assert d.a == 1 and d.missing is None                      # utils.py:35-37
    assert isinstance(d.test, AttrDict) and d.test.beam_size == 13
    assert d.test.set1.src_path == 'x' and d.test.nope is None
    assert [k for k in d.test if k.startswith('set')] == ['set1']


def test_shipped_configs_parse_and_carry_the_reference_keys():
    cdir = os.path.join(ROOT, 'configs')
    names = sorted(f for f in os.listdir(cdir) if f.endswith('.yaml'))
    assert len(names) >= 5
    vocab = {}
    for n in names:
        cfg = AttrDict(yaml.safe_load(open(os.path.join(cdir, n))))
        assert cfg.model == 'Transformer' and cfg.hidden_units == 1024 and cfg.num_heads == 16 and cfg.num_blocks == 6
        assert cfg.ff_activation == 'glu' and cfg.train.input_dim == 320 and cfg.test.max_target_length == 200
        vocab[n] = cfg.dst_vocab_size
        assert set(weights.variable_shapes(cfg)) == set(O.variable_shapes(O.make_config(dst_vocab_size=cfg.dst_vocab_size)))
    assert {3900, 1388, 28444, 3205} <= set(vocab.values())   # char / pinyin / word / SEAME


def test_expand_feed_dict_and_shards():
    X = np.arange(10)[:, None]
    fd = expand_feed_dict({('a', 'b', 'c', 'd'): X, 'k': 5})
    assert [len(fd[p]) for p in 'abcd'] == [3, 3, 2, 2] and fd['k'] == 5
    assert np.array_equal(np.concatenate([fd[p] for p in 'abcd']), X)
    assert shard_bounds(2, 4) == [(0, 1), (1, 2), (2, 2), (2, 2)]
    assert shard_bounds(10, 4) == O.expand_shards(10, 4)


def _write_vocab(path, n):
    with open(path, 'w', encoding='utf-8') as f:
        for i, w in enumerate(['<PAD>', '<UNK>', '<S>', '</S>'] + ['w%d' % i for i in range(n - 4)] + ['extra']):
            f.write('%s %d\n' % (w, 1000 - i))


def test_ark_roundtrip_and_datareader_batching(tmp_path):
    scp, arkf = str(tmp_path / 'feats.scp'), str(tmp_path / 'feats.ark')
    rng = np.random.default_rng(0)
    feats = [rng.standard_normal((t, 320)).astype(np.float32) for t in (55, 61, 120, 58, 49)]
    w = ark.ArkWriter(scp)
    for i, f in enumerate(feats):
        w.write_next_utt(arkf, 'utt%d' % i, f)
    w.close()
    r = ark.ArkReader(scp)
    assert len(r) == 5
    for i, f in enumerate(feats):
        uid, m, looped = r.read_next_utt()
        assert uid == 'utt%d' % i and not looped and np.array_equal(m, f)
    assert r.read_next_utt()[2] is True                           # loops after the last entry (ark.py:164-166)
    assert np.array_equal(r.read_utt('utt3'), feats[3])

    vocab = str(tmp_path / 'vocab.txt')
    _write_vocab(vocab, 20)
    cfg = product_config(tiny_config(dst_vocab_size=20))
    cfg['dst_vocab'] = vocab
    dr = DataReader(cfg)
    assert dr.idx2dst[3] == '</S>' and len(dr.idx2dst) == 20
    batches = list(dr.get_test_batches(scp, 2))
    assert [b[1] for b in batches] == [['utt0', 'utt1'], ['utt2', 'utt3'], ['utt4']]
    X = batches[0][0]
    assert X.shape == (2, 61, 320) and X.dtype == np.float32 and np.all(X[0, 55:] == 0) and np.array_equal(X[1], feats[1])
    bb = list(dr.get_test_batches_with_buckets(scp, 100))
    # buckets: 55,58 -> bucket 60 (emitted once > 100 frames); 61 -> 70; 120 -> 130 (emitted at once); 49 -> 50
    assert [b[1] for b in bb] == [['utt2'], ['utt0', 'utt3'], ['utt4'], ['utt1']]
    assert sorted(sum([b[1] for b in bb], [])) == ['utt%d' % i for i in range(5)]
    assert dr.indices_to_words(np.array([[5, 6, 3, 7], [3, 3, 3, 3], [4, 4, 4, 4]])) == ['w1 w2', '', 'w0 w0 w0 w0']


def test_target_batches_for_ppl(tmp_path):
    """get_test_batches_with_target / _create_target_batch (evaluate.py:76-87; utils.py:108-116,222-250)."""
    scp, arkf = str(tmp_path / 'feats.scp'), str(tmp_path / 'feats.ark')
    rng = np.random.default_rng(1)
    feats = [rng.standard_normal((t, 320)).astype(np.float32) for t in (12, 20, 15)]
    w = ark.ArkWriter(scp)
    for i, f in enumerate(feats):
        w.write_next_utt(arkf, 'utt%d' % i, f)
    w.close()
    vocab = str(tmp_path / 'vocab.txt')
    _write_vocab(vocab, 20)
    text = str(tmp_path / 'text')
    with open(text, 'w') as f:
        f.write('utt0 w1 w2 w3\n\nutt2   w0  nosuchword\n')          # utt1 has no target; blank line ignored
    cfg = product_config(tiny_config(dst_vocab_size=20))
    cfg['dst_vocab'] = vocab
    dr = DataReader(cfg)
    Y, mask = dr._create_target_batch([['w1', 'w2'], ['w0']], dr.dst2idx)
    assert Y.tolist() == [[5, 6, 3], [4, 3, 0]] and mask.tolist() == [[0, 0, 0], [0, 0, 1]] and Y.dtype == np.int32
    batches = list(dr.get_test_batches_with_target(scp, text, 8))
    assert len(batches) == 1
    X, Yb = batches[0]
    assert X.shape == (2, 15, 320) and np.array_equal(X[0, :12], feats[0]) and np.array_equal(X[1], feats[2])
    assert Yb.tolist() == [[5, 6, 7, 3], [4, 1, 3, 0]]                  # unknown word -> <UNK> = 1, </S> appended, PAD 0


def test_ark_double_and_compressed(tmp_path):
    import struct
    p = str(tmp_path / 'x.ark')
    m = np.arange(6, dtype=np.float64).reshape(2, 3)
    with open(p, 'wb') as f:
        f.write(b'a ')
        off1 = f.tell()
        f.write(b'\0BDM ' + struct.pack('<bi', 4, 2) + struct.pack('<bi', 4, 3) + m.tobytes())
        f.write(b'b ')
        off2 = f.tell()
        # BCM2: 16-bit, value = raw * range / 65535 + min
        raw = np.array([[0, 100], [-200, 32767]], dtype='<i2')
        f.write(b'\0BCM2 ' + struct.pack('<ffii', -1.0, 4.0, 2, 2) + raw.tobytes())
    with open(str(tmp_path / 'x.scp'), 'w') as f:
        f.write('a %s:%d\nb %s:%d\n' % (p, off1, p, off2))
    r = ark.ArkReader(str(tmp_path / 'x.scp'))
    assert np.array_equal(r.read_utt('a'), m.astype(np.float32))
    assert np.allclose(r.read_utt('b'), raw.astype(np.float32) * 4.0 / 65535.0 - 1.0)


def test_random_init_follows_q11_and_checkpoint_io(tmp_path):
    cfg = product_config(tiny_config())
    W = weights.random_init(cfg, seed=3)
    shapes = weights.variable_shapes(cfg)
    assert set(W) == set(shapes)
    k = W['encoder/block_0/encoder_self_attention/qkv_transform/kernel']
    lim = np.sqrt(6.0 / (k.shape[0] + k.shape[1]))
    assert k.dtype == np.float32 and np.abs(k).max() <= lim and np.abs(k).max() > 0.9 * lim
    assert np.all(W['encoder/LayerNorm/gamma'] == 1) and np.all(W['encoder/src_change/bias'] == 0)
    d = tmp_path / 'model_dir'
    d.mkdir()
    extra = dict(W)
    extra['global_step'] = np.array(7, np.int64)
    extra['encoder/src_change/kernel/Adam'] = np.zeros_like(W['encoder/src_change/kernel'])
    extra['beta1_power'] = np.float32(0.5)
    weights.save_npz(str(d / 'averaged.ckpt-0'), extra)
    (d / 'checkpoint').write_text('model_checkpoint_path: "averaged.ckpt-0"\nall_model_checkpoint_paths: "averaged.ckpt-0"\n')
    path = weights.latest_checkpoint(str(d))
    assert path.endswith('averaged.ckpt-0')
    got = weights.load_checkpoint(path)
    assert set(got) == set(W) and all(np.array_equal(got[k], W[k]) for k in W)


def test_c_abi_library_exports_every_declared_symbol():
    """The shared library loads without a GPU and exports every function include/asr_amd.h declares."""
    hdr = open(os.path.join(ROOT, 'include', 'asr_amd.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    names = set(re.findall(r'\b(asr_[a-z_0-9]+)\s*\(', hdr))
    assert len(names) >= 20 and 'asr_decode' in names and 'asr_op_dense' in names
    from asr_amd import _lib
    lib = C.CDLL(_lib.LIB_PATH)
    for n in sorted(names):
        assert hasattr(lib, n), 'missing export %s' % n
    # and the ctypes binding covers all of them
    src = open(os.path.join(ROOT, 'asr_amd', '_lib.py')).read()
    for n in names:
        assert "'%s'" % n in src, 'asr_amd/_lib.py does not bind %s' % n


def test_argument_errors_without_gpu():
    """asr_create must fail loudly (not fall back) when no GPU is visible or the config is unsupported."""
    from asr_amd import _lib
    from asr_amd.model import Engine
    bad = product_config(tiny_config(hidden_units=96, num_heads=2))      # d_head != 64
    with pytest.raises(_lib.AsrError):
        Engine(bad, 0)
    if _lib.lib.asr_device_count() == 0:
        with pytest.raises(_lib.AsrError) as ei:
            Engine(product_config(tiny_config()), 0)
        assert 'HIP device' in str(ei.value)


def test_product_path_never_imports_the_oracle():
    for root, _, files in os.walk(os.path.join(ROOT, 'asr_amd')):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(root, f)).read()
                assert 'oracle' not in re.sub(r'#.*', '', src).replace('asr_oracle', 'oracle') or \
                    not re.search(r'^\s*(from|import)\s+oracle', src, flags=re.M), f
                assert not re.search(r'^\s*(from|import)\s+oracle', src, flags=re.M), f


def test_engine_pool_map_keeps_order_bounds_what_is_in_flight_and_propagates_errors():
    """asr_amd.model.EnginePool.map with stand-in handles (no GPU): results in item order, at most one item per handle
    in flight, items drawn lazily, a failing call re-raised at its place in the order."""
    import threading
    import time
    from asr_amd.model import EnginePool

    class Fake(object):
        def __init__(self, tag):
            self.tag = tag
    pool = object.__new__(EnginePool)
    pool.engines = [Fake(0), Fake(1), Fake(2)]
    lock = threading.Lock()
    state = {'now': 0, 'peak': 0, 'drawn': 0}
    busy = set()

    def call(e, item):
        with lock:
            assert e.tag not in busy                         # a handle serves one call at a time
            busy.add(e.tag)
            state['now'] += 1
            state['peak'] = max(state['peak'], state['now'])
        time.sleep(0.002 * (5 - item % 5))                   # later items may finish first
        with lock:
            busy.discard(e.tag)
            state['now'] -= 1
        if item == 17:
            raise ValueError('item 17')
        return item * item

    def items(n):
        for i in range(n):
            state['drawn'] = i + 1
            yield i
    got = []
    for r in pool.map(call, items(12)):
        got.append(r)
        assert state['drawn'] <= len(got) + len(pool.engines)     # lazily: never further ahead than the handles can hold
    assert got == [i * i for i in range(12)]
    assert 2 <= state['peak'] <= 3
    out = []
    with pytest.raises(ValueError, match='item 17'):
        for r in pool.map(call, items(30)):
            out.append(r)
    assert out == [i * i for i in range(17)]
    # one handle: plain sequential calls on the caller's thread
    pool.engines = [Fake(0)]
    assert list(pool.map(lambda e, it: (threading.get_ident(), it), range(3))) == [(threading.get_ident(), i) for i in range(3)]
    # regimes: inside map (and inside together(), nestable) every handle is in the shared-GPU regime, outside in the regime of a
    # handle that has the GPU to itself - also after a call failed
    log = []

    class Handle(Fake):
        _h = 1

        def set_sharing(self, shared, sub):
            log.append((self.tag, shared, sub))
    pool = object.__new__(EnginePool)
    pool.engines = [Handle(0), Handle(1)]
    pool._alone, pool._together, pool._regime = (0, 0), (1, 1), (0, 0)
    pool._regime_lock, pool._regime_users = threading.Lock(), 0
    seen = []

    def probe(e, item):
        seen.append(pool._regime)
        if item == 3:
            raise ValueError('item 3')
        return item
    assert list(pool.map(probe, range(3))) == [0, 1, 2]
    assert set(seen) == {(1, 1)} and pool._regime == (0, 0)
    assert log == [(0, 1, 1), (1, 1, 1), (0, 0, 0), (1, 0, 0)]
    with pytest.raises(ValueError):
        list(pool.map(probe, range(6)))
    assert pool._regime == (0, 0) and pool._regime_users == 0
    log[:] = []
    with pool.together():
        with pool.together():
            assert pool._regime == (1, 1)
        assert pool._regime == (1, 1) and len(log) == 2
    assert pool._regime == (0, 0) and len(log) == 4


def test_predict_batches_merge_utts_groups_consecutive_batches_and_keeps_their_order():
    """`test.merge_utts` (build extension): Model.predict_batches joins CONSECUTIVE batches into joint decodes of at most that
    many utterances, hands the results back batch by batch in order, decodes a lone batch (and an empty one) with the plain
    call, and leaves everything alone when the key is absent.  Stand-in handle, no GPU."""
    from asr_amd.model import Model
    calls = []

    class Fake(object):
        def decode(self, X):
            calls.append(('one', X.shape[0]))
            return np.full((X.shape[0], 2), X.shape[1], np.int32) if X.shape[0] else np.zeros((0, 0), np.int32)

        def decode_groups(self, Xs):
            calls.append(('joint', [X.shape[0] for X in Xs]))
            return [np.full((X.shape[0], 2), X.shape[1], np.int32) for X in Xs]

    def batches():
        for n, t in [(2, 10), (2, 11), (1, 12), (0, 5), (3, 13), (3, 14), (5, 15)]:
            yield np.zeros((n, t, 4), np.float32)

    cfg = product_config(tiny_config())
    cfg.test['merge_utts'] = 4
    m = Model(cfg, 1, devices=[0])
    m.engines = [Fake()]
    out = list(m.predict_batches(batches()))
    assert [y.shape[0] for y in out] == [2, 2, 1, 0, 3, 3, 5]
    assert [int(y[0, 0]) for y in out if y.shape[0]] == [10, 11, 12, 13, 14, 15]
    assert calls == [('joint', [2, 2]), ('one', 1), ('one', 0), ('one', 3), ('one', 3), ('one', 5)]
    calls[:] = []
    cfg.test['merge_utts'] = None
    m.predict = lambda X: ('plain', X.shape[0])
    assert [r for r in m.predict_batches(batches())][:2] == [('plain', 2), ('plain', 2)] and calls == []


def test_hw_queue_policy(monkeypatch):
    """asr_amd/hwqueues.py: ONE policy for bench.py and the product path - a value below the eight queues a four-handle pool
    wants is raised (unset or set), a higher one is kept, ASR_AMD_KEEP_HW_QUEUES=1 keeps anything, and once the GPU is in
    use nothing is changed and the caller is told (`late`)."""
    from asr_amd import hwqueues
    monkeypatch.setattr(hwqueues, '_gpu_started', [False])
    monkeypatch.delenv('ASR_AMD_KEEP_HW_QUEUES', raising=False)
    monkeypatch.delenv('GPU_MAX_HW_QUEUES', raising=False)
    r = hwqueues.ensure()
    assert r['value'] == 8 and not r['late'] and os.environ['GPU_MAX_HW_QUEUES'] == '8' and 'unset' in r['source']
    monkeypatch.setenv('GPU_MAX_HW_QUEUES', '4')
    r = hwqueues.ensure()
    assert r['value'] == 8 and os.environ['GPU_MAX_HW_QUEUES'] == '8' and 'raised from 4' in r['source']
    monkeypatch.setenv('GPU_MAX_HW_QUEUES', '16')
    r = hwqueues.ensure()
    assert r['value'] == 16 and os.environ['GPU_MAX_HW_QUEUES'] == '16' and 'kept' in r['source']
    monkeypatch.setenv('GPU_MAX_HW_QUEUES', '4')
    monkeypatch.setenv('ASR_AMD_KEEP_HW_QUEUES', '1')
    r = hwqueues.ensure()
    assert r['value'] == 4 and os.environ['GPU_MAX_HW_QUEUES'] == '4' and not r['late']
    monkeypatch.delenv('ASR_AMD_KEEP_HW_QUEUES')
    monkeypatch.setattr(hwqueues, '_gpu_started', [True])              # the process already uses the GPU: too late
    r = hwqueues.ensure()
    assert r['value'] == 4 and r['late'] and os.environ['GPU_MAX_HW_QUEUES'] == '4'
    monkeypatch.delenv('GPU_MAX_HW_QUEUES')
    assert hwqueues.ensure()['value'] == 4 and hwqueues.current() == 4


def test_model_defaults_to_the_measured_handle_count(monkeypatch):
    """With `test.engines_per_gpu` unset and one tower, Model takes four handles when the process has (or can still get) eight
    hardware queues and three otherwise; an explicit key or ASR_AMD_ENGINES wins; several towers keep one handle per GPU."""
    from asr_amd import hwqueues
    from asr_amd.model import Model
    ocfg = tiny_config()
    monkeypatch.delenv('ASR_AMD_ENGINES', raising=False)
    monkeypatch.delenv('ASR_AMD_KEEP_HW_QUEUES', raising=False)
    monkeypatch.delenv('GPU_MAX_HW_QUEUES', raising=False)
    monkeypatch.setenv('ASR_AMD_DEVICE', '0')                           # (no device query on this CPU-only box)
    monkeypatch.setattr(hwqueues, '_gpu_started', [False])
    assert Model(product_config(ocfg), 1)._engines_per_gpu == 4 and os.environ['GPU_MAX_HW_QUEUES'] == '8'
    monkeypatch.setenv('GPU_MAX_HW_QUEUES', '4')
    monkeypatch.setattr(hwqueues, '_gpu_started', [True])
    assert Model(product_config(ocfg), 1)._engines_per_gpu == 3 and os.environ['GPU_MAX_HW_QUEUES'] == '4'
    cfg = product_config(ocfg)
    cfg.test['engines_per_gpu'] = 2
    assert Model(cfg, 1)._engines_per_gpu == 2
    monkeypatch.setenv('ASR_AMD_ENGINES', '1')
    assert Model(product_config(ocfg), 1)._engines_per_gpu == 1
    monkeypatch.delenv('ASR_AMD_ENGINES')
    # several towers: on ONE GPU (this process's) the same default - their shards are decoded jointly and batches go over the
    # pool; towers on GPUs of their own keep one handle per GPU
    assert Model(product_config(ocfg, num_gpus=3), 3)._engines_per_gpu == 3
    assert Model(product_config(ocfg, num_gpus=3), 3, devices=[0, 1, 2])._engines_per_gpu == 1


def test_towers_on_one_gpu_are_one_joint_decode_and_batches_go_over_the_pool():
    """`test.num_gpus: 4` (what the reference's YAMLs ship) in a process with one GPU: every batch is split into its towers'
    shards exactly as expand_feed_dict does (utils.py:420-443), the shards of one GPU are ONE decode_groups call, empty shards
    give no rows (model.py:184-187), predictions are padded with 3 to the longest and concatenated in tower order
    (model.py:192-201); predict_batches keeps several batches in flight over the pool and returns them in order."""
    from asr_amd.model import EnginePool, Model
    calls = []

    class Fake(object):
        _h = 1

        def decode(self, X):
            calls.append(('decode', [X.shape[0]]))
            return np.full((X.shape[0], 2), int(X[0, 0, 0]), np.int32)

        def decode_groups(self, Xs):
            calls.append(('groups', [X.shape[0] for X in Xs]))
            # group g answers with g + 2 columns of its first feature value
            return [np.full((X.shape[0], 2 + g), int(X[0, 0, 0]), np.int32) for g, X in enumerate(Xs)]

        def set_sharing(self, *a):
            pass

    cfg = product_config(tiny_config(beam_size=2), num_gpus=4)
    m = Model(cfg, 4, devices=[0])
    pool = object.__new__(EnginePool)
    pool.engines = [Fake(), Fake()]
    m.pools, m.engines = [pool], [pool.engines[0]]
    X = np.zeros((10, 5, 4), np.float32)
    X[:, 0, 0] = np.arange(10) + 10                       # utterance u carries 10 + u
    Y = m.predict(X)                                      # shards 3, 3, 2, 2 (the first N % n shards get one more)
    assert calls == [('groups', [3, 3, 2, 2])]
    assert Y.shape == (10, 5) and [int(v) for v in Y[:, 0]] == [10] * 3 + [13] * 3 + [16] * 2 + [18] * 2
    assert np.all(Y[0:3, 2:] == 3) and np.all(Y[3:6, 3:] == 3) and np.all(Y[8:10, :5] == 18)       # padded with 3 to the longest
    calls[:] = []
    Y = m.predict(X[:2])                                  # shards 1, 1, 0, 0: the empty towers give no rows
    assert calls == [('groups', [1, 1])] and Y.shape[0] == 2
    calls[:] = []
    Y = m.predict(X[:1])                                  # one shard with rows: a plain decode
    assert calls == [('decode', [1])] and Y.shape == (1, 2)
    calls[:] = []
    out = list(m.predict_batches(iter([X, X[:6], X[:3]])))
    assert [y.shape[0] for y in out] == [10, 6, 3]
    assert sorted(c[1] for c in calls) == sorted([[3, 3, 2, 2], [2, 2, 1, 1], [1, 1, 1]])


def test_predict_batches_pairs_whole_batches_and_keeps_their_order():
    """Without `test.merge_utts`, a pool of several handles decodes consecutive WHOLE batches (>= 448 beam rows each) two per
    call and everything else one per call; results come back batch by batch in order; `test.pair_batches: false` turns it off."""
    from asr_amd.model import EnginePool, Model
    calls = []

    class Fake(object):
        _h = 1

        def decode(self, X):
            calls.append([X.shape[0]])
            return np.full((X.shape[0], 2), X.shape[1], np.int32)

        def decode_groups(self, Xs):
            calls.append([X.shape[0] for X in Xs])
            return [np.full((X.shape[0], 2), X.shape[1], np.int32) for X in Xs]

        def set_sharing(self, *a):
            pass

    def batches():
        for n, t in [(64, 10), (64, 11), (64, 12), (5, 13), (64, 14), (60, 15), (44, 16), (64, 17), (45, 18), (50, 19)]:
            yield np.zeros((n, t, 4), np.float32)
    for pair in (None, False):
        cfg = product_config(tiny_config(beam_size=10))
        if pair is not None:
            cfg.test['pair_batches'] = pair
        m = Model(cfg, 1, devices=[0])
        pool = object.__new__(EnginePool)
        pool.engines = [Fake(), Fake()]
        m.pools, m.engines = [pool], [pool.engines[0]]
        calls[:] = []
        out = list(m.predict_batches(batches()))
        assert [int(y[0, 0]) for y in out] == [10, 11, 12, 13, 14, 15, 16, 17, 18, 19]
        assert [y.shape[0] for y in out] == [64, 64, 64, 5, 64, 60, 44, 64, 45, 50]
        if pair is None:
            assert sorted(calls) == sorted([[64, 64], [64], [5], [64, 60], [44], [64, 45], [50]])
        else:
            assert sorted(calls) == sorted([[64], [64], [64], [5], [64], [60], [44], [64], [45], [50]])


def test_pairing_threshold_matches_the_engine():
    """asr_amd.model pairs batches from the row count at which the engine's shared-GPU K-slice rule stops following the number of
    rows (engine.hip WHOLE_BATCH_ROWS_DEFAULT): the two constants must stay one number, or a pair could differ from its halves."""
    import re
    from asr_amd import model
    src = open(os.path.join(os.path.dirname(model.__file__), 'csrc', 'engine.hip')).read()
    m = re.search(r'constexpr int WHOLE_BATCH_ROWS_DEFAULT = (\d+);', src)
    assert m and int(m.group(1)) == model.PAIR_MIN_ROWS
