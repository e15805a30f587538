"""GPU tests of what round 3 put around the engine handles: one weight store per GPU (asr_share_weights), the two regimes of
a pool's handles (asr_set_sharing), the default handle count of the drop-in, and `python -m asr_amd.evaluate` as one process
per GPU (two ranks rehearsed on the one GPU of the test box, gloo for the gather) against the single-process file."""
import os
import re
import socket
import subprocess
import sys

import numpy as np
import pytest
import yaml

from oracle import asr_oracle as O
from tests.helpers import product_config, tiny_config, tiny_inputs
from tests.test_gpu_configs import _write_vocab_bpe

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_handles_of_a_pool_share_one_weight_store():
    """EnginePool.load_weights packs and uploads once; the other handles adopt the first one's device copy.  Same hypotheses
    from every handle, the pool's footprint holds the weights once, either handle may be destroyed first, and handles that
    do not match are refused."""
    from asr_amd._lib import AsrError
    from asr_amd.model import Engine, EnginePool
    ocfg = tiny_config(beam_size=3, max_target_length=12)
    W = O.random_weights(ocfg, seed=1, bias_scale=0.1)
    X = tiny_inputs((60, 90, 75), seed=2)
    want = O.build_test_model(X, W, ocfg, num_gpus=1, algo='kv')
    lone = Engine(product_config(ocfg), 0)
    lone.load_weights(W)
    weights_bytes = lone.device_bytes()                       # nothing but the weights and the timing table so far
    pool = EnginePool(product_config(ocfg), 0, 4)
    pool.load_weights(W)
    try:
        per = [e.device_bytes() for e in pool.engines]
        assert per[0] >= weights_bytes * 0.99
        assert all(b < 0.05 * per[0] + (1 << 20) for b in per[1:])           # the others hold no copy of their own
        assert pool.device_bytes() < 1.2 * weights_bytes + (4 << 20)
        for e in pool.engines:
            assert np.array_equal(e.decode(X), want)
        assert [np.array_equal(Y, want) for Y in pool.map(lambda e, x: e.decode(x), [X] * 6)] == [True] * 6
        # the store outlives the handle that filled it
        pool.engines[0].close()
        assert np.array_equal(pool.engines[1].decode(X), want) and np.array_equal(pool.engines[3].decode(X), want)
        # refused: another model shape, another precision, a handle that already has weights, an unfinalized source
        other = Engine(product_config(tiny_config(beam_size=3, max_target_length=12, dst_vocab_size=60)), 0)
        with pytest.raises(AsrError, match='different model shapes'):
            other.share_weights(lone)
        other.close()
        other = Engine(product_config(ocfg, precision='f32'), 0)
        with pytest.raises(AsrError, match='different model shapes or precision'):
            other.share_weights(lone)
        empty = Engine(product_config(ocfg), 0)
        with pytest.raises(AsrError, match='no finalized weights'):
            other.share_weights(empty)
        other.close()
        with pytest.raises(AsrError, match='already finalized'):
            lone.share_weights(pool.engines[1])
        # beam size and length limits are per handle, not per store
        wide = Engine(product_config(tiny_config(beam_size=5, max_target_length=9)), 0)
        wide.share_weights(lone)
        ocfg5 = tiny_config(beam_size=5, max_target_length=9)
        assert np.array_equal(wide.decode(X), O.build_test_model(X, W, ocfg5, num_gpus=1, algo='kv'))
        wide.close()
        empty.close()
    finally:
        pool.close()
        lone.close()


def test_pool_regimes_alone_and_together_give_the_oracles_hypotheses():
    """A handle of a pool used alone (Model.predict) runs in the regime of a handle that has the GPU to itself; inside
    EnginePool.map / together() all handles run in the shared-GPU regime (fewer K-slices from 448 beam rows, one run of
    kernels per batch).  Full-size char architecture, 64 short utterances x beam 10 = 640 rows, 5 steps: the float32
    oracle's hypotheses in both regimes, and asr_set_sharing refuses values outside its ranges."""
    from asr_amd._lib import AsrError
    from asr_amd.model import EnginePool
    ocfg = O.make_config(beam_size=10, max_target_length=4)
    W = O.random_weights(ocfg, seed=51)
    X = O.make_batch(O.synthetic_fbank(64, [90 + 3 * (i % 7) for i in range(64)], seed=52))
    want = O.build_test_model(X, W, ocfg, num_gpus=1, algo='kv')
    pool = EnginePool(product_config(ocfg), 0, 2)
    pool.load_weights(W)
    try:
        assert pool._alone == (0, 0) and pool._together == (1, 1)
        assert np.array_equal(pool.engines[0].decode(X), want)                      # alone: two half batches, chip-filling K-slices
        got = list(pool.map(lambda e, x: e.decode(x), [X, X, X]))
        assert all(np.array_equal(Y, want) for Y in got)                            # together
        assert pool._regime == (0, 0)
        assert np.array_equal(pool.engines[1].decode(X), want)
        with pytest.raises(AsrError):
            pool.engines[0].set_sharing(2, 0)
        with pytest.raises(AsrError):
            pool.engines[0].set_sharing(0, 5)
    finally:
        pool.close()


def _evaluate_case(tmp_path, n_utts=9):
    from asr_amd import ark, tf_bundle
    ocfg = tiny_config(beam_size=3, max_target_length=14)
    W = O.random_weights(ocfg, seed=86, bias_scale=0.1)
    model_dir = tmp_path / 'exp'
    model_dir.mkdir()
    tf_bundle.write_bundle(str(model_dir / 'model_step_900'), dict(W))
    tf_bundle.write_checkpoint_state(str(model_dir), 'model_step_900')
    vocab_path = str(tmp_path / 'vocab.txt')
    words = _write_vocab_bpe(vocab_path, ocfg.dst_vocab_size)
    lens = [180, 270, 120, 225, 99, 150, 240, 201, 132][:n_utts]
    feats = [O.frame_stack(r) for r in O.synthetic_fbank(len(lens), lens, seed=62)]
    scp, arkf = str(tmp_path / 'feats.scp'), str(tmp_path / 'feats.ark')
    w = ark.ArkWriter(scp)
    for i, f in enumerate(feats):
        w.write_next_utt(arkf, 'spk%02d-utt%d' % (i % 2, i), f)
    w.close()

    def config(out_path, **test_extra):
        cfg = {
            'model': 'Transformer', 'dst_vocab': vocab_path, 'dst_vocab_size': ocfg.dst_vocab_size,
            'hidden_units': ocfg.hidden_units, 'scale_embedding': True, 'tie_embedding_and_softmax': False,
            'num_blocks': ocfg.num_blocks, 'num_heads': ocfg.num_heads, 'ff_activation': 'glu', 'model_dir': str(model_dir),
            'train': {'input_dim': 320},
            'test': {'batch_size': 2, 'max_target_length': 14, 'lp_alpha': 0.6, 'beam_size': 3, 'num_gpus': 1,
                     'set1': {'src_path': scp, 'ref_path': 'unused', 'output_path': out_path, 'cmd': None}},
        }
        cfg['test'].update(test_extra)
        path = out_path + '.yaml'
        with open(path, 'w') as f:
            yaml.safe_dump(cfg, f)
        return path
    return ocfg, W, feats, words, config


@pytest.mark.parametrize('bucketed', [False, True])
def test_evaluate_as_one_process_per_gpu_writes_the_single_process_file(tmp_path, bucketed):
    """`python -m asr_amd.evaluate -c cfg` under WORLD_SIZE = 2 (two ranks on this box's one GPU, gloo for the gather; on a
    node: torch.distributed.run with the nccl backend = RCCL): the batches of the single-process run dealt over the ranks,
    one all_gather, rank 0 writes - byte for byte the file the single-process run writes, which in turn is the oracle's
    hypotheses on the same batches through indices_to_words and the BPE strip."""
    ocfg, W, feats, words, config = _evaluate_case(tmp_path)
    extra = {'bucketed_reader': True, 'tokens_per_batch': 150} if bucketed else {}
    single_out, sharded_out = str(tmp_path / 'single.txt'), str(tmp_path / 'sharded.txt')
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT', 'MASTER_ADDR')}
    env['PYTHONPATH'] = ROOT + os.pathsep + env.get('PYTHONPATH', '')
    r = subprocess.run([sys.executable, '-m', 'asr_amd.evaluate', '-c', config(single_out, **extra)], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cfg2 = config(sharded_out, **extra)
    procs = []
    for rank in range(2):
        e = dict(env, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                 ASR_AMD_DIST_BACKEND='gloo', ASR_AMD_DEVICE='0')
        procs.append(subprocess.Popen([sys.executable, '-m', 'asr_amd.evaluate', '-c', cfg2], env=e, cwd=ROOT,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert [p.returncode for p in procs] == [0, 0], outs
    got = open(sharded_out, 'rb').read()
    assert got == open(single_out, 'rb').read()
    assert not os.path.exists(sharded_out + '.tmp')
    # ... and that file is the oracle's: the reader's batches, each decoded on its own
    from asr_amd import dist as D
    lengths = [len(f) for f in feats]
    batches = D.bucket_batches(lengths, 150) if bucketed else [list(range(b, min(b + 2, len(feats)))) for b in range(0, len(feats), 2)]
    idx2 = {i: t for i, t in enumerate(words)}
    lines = []
    for idx in batches:
        tmax = max(lengths[i] for i in idx)
        Xb = np.zeros((len(idx), tmax, 320), np.float32)
        for r_, i in enumerate(idx):
            Xb[r_, :lengths[i]] = feats[i]
        Yb = O.build_test_model(Xb, W, ocfg, num_gpus=1, algo='kv')
        for i, sent in zip(idx, O.indices_to_words(Yb, idx2)):
            lines.append('spk%02d-utt%d' % (i % 2, i) + '\t' + sent)
    want = ''.join(re.sub(r'(@@ )|(@@ ?$)', '', ln) + '\n' for ln in lines)
    assert got.decode('utf-8') == want


def test_whole_batches_are_decoded_in_pairs_with_bit_identical_hypotheses():
    """Model.predict_batches with several handles: consecutive batches of at least 448 beam rows go two per library call
    (Engine.decode_groups; one pass over every weight plane for both).  From that row count on a pool's handles pick their
    K-slices independently of the number of rows, so the pair must reproduce the one-batch-per-call hypotheses BIT for bit:
    full-size char architecture, batches of 64 / 60 / 50 / 64 / 20 utterances with different padded lengths, beam 10, and the
    float32 oracle's hypotheses for every batch."""
    from asr_amd.model import Model
    ocfg = O.make_config(beam_size=10, max_target_length=4)
    W = O.random_weights(ocfg, seed=51)
    shapes = [(64, 90), (60, 120), (50, 75), (64, 81), (20, 60)]
    Xs = [O.make_batch(O.synthetic_fbank(n, [top - 3 * (i % 5) for i in range(n)], seed=300 + j)) for j, (n, top) in enumerate(shapes)]
    want = [O.build_test_model(X, W, ocfg, num_gpus=1, algo='kv') for X in Xs]
    got = {}
    for pair in (True, False):
        cfg = product_config(ocfg)
        cfg.test['engines_per_gpu'] = 2
        cfg.test['pair_batches'] = pair
        m = Model(cfg, 1, devices=[0])
        m.build_test_model()
        m.restore(W)
        calls = []
        for e in m.pools[0].engines:
            dg, dc = e.decode_groups, e.decode
            e.decode_groups = lambda Xg, dg=dg: (calls.append([X.shape[0] for X in Xg]), dg(Xg))[1]
            e.decode = lambda X, dc=dc: (calls.append([X.shape[0]]), dc(X))[1]
        try:
            got[pair] = list(m.predict_batches(iter(Xs)))
        finally:
            m.close()
        assert sorted(calls) == ([[20], [50, 64], [64, 60]] if pair else [[20], [50], [60], [64], [64]])
    for Yp, Y1, Yw in zip(got[True], got[False], want):
        assert Yp.shape == Y1.shape and np.array_equal(Yp, Y1)          # paired == alone, bit for bit
        assert np.array_equal(Y1, Yw)                                    # == the oracle's


def test_the_shipped_test_section_four_towers_on_one_gpu():
    """`test.num_gpus: 4`, `beam_size: 13` (the test section the reference's char YAML ships, config_template_char_...yaml:49-54)
    in a process with one GPU: every batch is split into four shards as expand_feed_dict does (utils.py:420-443), the four
    shards are ONE joint decode (676 beam rows here), predictions are padded with 3 and concatenated in tower order
    (model.py:192-201).  Full-size char architecture; hypotheses identical to the oracle's four-tower evaluation, and
    predict_batches (several batches in flight over the default pool) returns what predict gives batch by batch; with
    `test.joint_towers: false` the towers are decoded one after the other and give the same."""
    from asr_amd.model import Model
    ocfg = O.make_config(beam_size=13, max_target_length=4)
    W = O.random_weights(ocfg, seed=61)
    Xs = [O.make_batch(O.synthetic_fbank(n, [top - 3 * (i % 6) for i in range(n)], seed=400 + j)) for j, (n, top) in enumerate([(52, 90), (50, 75), (3, 60)])]
    want = [O.build_test_model(X, W, ocfg, num_gpus=4, algo='kv') for X in Xs]
    for joint in (True, False):
        cfg = product_config(ocfg, 4)
        cfg.test['joint_towers'] = joint
        m = Model(cfg, 4, devices=[0])
        assert m._engines_per_gpu in (3, 4)                                # the default pool, although the YAML names four towers
        m.build_test_model()
        m.restore(W)
        calls = []
        for e in m.pools[0].engines:
            dg, dc = e.decode_groups, e.decode
            e.decode_groups = lambda Xg, dg=dg: (calls.append([X.shape[0] for X in Xg]), dg(Xg))[1]
            e.decode = lambda X, dc=dc: (calls.append([X.shape[0]]), dc(X))[1]
        try:
            one = [m.predict(X) for X in Xs]
            n1 = len(calls)
            many = list(m.predict_batches(iter(Xs)))
        finally:
            m.close()
        if joint:
            assert calls[:n1] == [[13, 13, 13, 13], [13, 13, 12, 12], [1, 1, 1]]
        else:
            assert calls[:n1] == [[13]] * 4 + [[13], [13], [12], [12]] + [[1]] * 3
        for Y1, Ym, Yw in zip(one, many, want):
            assert np.array_equal(Y1, Yw)
            assert np.array_equal(Ym, Yw)


def test_calls_of_one_two_and_three_whole_batches_give_bit_identical_hypotheses():
    """bench.py decodes two 64-utterance batches per library call and three when a handle is left with an odd last step
    (plan_calls).  Utterances are independent and, from 448 beam rows up, a handle that shares the GPU picks its K-slices
    whatever the number of rows: 64, 128 and 192 utterances of one padded length in one call must give every utterance the
    hypothesis its own 64-utterance batch gives, bit for bit - and the float32 oracle's."""
    from asr_amd.model import Engine
    ocfg = O.make_config(beam_size=10, max_target_length=4)
    W = O.random_weights(ocfg, seed=51)
    X = O.make_batch(O.synthetic_fbank(192, [96 - 3 * (i % 7) for i in range(192)], seed=77))
    cfg = product_config(ocfg, sub_batches=1)
    cfg.test['shared_gpu'] = 1
    eng = Engine(cfg, 0)
    eng.load_weights(W)
    try:
        alone = [eng.decode(X[i:i + 64]) for i in (0, 64, 128)]
        two = eng.decode(X[:128])
        three = eng.decode(X)
        assert np.array_equal(two, np.concatenate(alone[:2]))
        assert np.array_equal(three, np.concatenate(alone))
        assert np.array_equal(alone[0], O.build_test_model(X[:64], W, ocfg, num_gpus=1, algo='kv'))
    finally:
        eng.close()
