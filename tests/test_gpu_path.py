"""End-to-end parity of the HIP path (through the C ABI) against the CPU oracle.

Bars (BASELINE.json north_star): identical greedy and beam token sequences on the fixed synthetic
sets below; logits within 1e-3 relative of the oracle (measured against its float64 evaluation; the
observed error is ~1e-5).  "Relative" is read as max-abs difference / max-abs reference, over the whole
[rows, vocab] tensor (tests.helpers.rel_err) AND per hypothesis row (rel_err_rows); a per-element ratio
has no meaning where a logit crosses zero.  PARITY UNPINNED by the reference itself (it has no fixtures and cannot
run here) -- see oracle/asr_oracle.py.
"""
import numpy as np
import pytest

from oracle import asr_oracle as O
from tests.helpers import product_config, rel_err, rel_err_rows, tiny_config, tiny_inputs

pytestmark = pytest.mark.gpu

LOGIT_TOL = 1e-3


def _engine(ocfg, W, device=0, sub_batches=None):
    from asr_amd.model import Engine
    e = Engine(product_config(ocfg, sub_batches=sub_batches), device)
    e.load_weights(W)
    return e


def _oracle_decode(X, W, ocfg, dtype=np.float32, algo='kv'):
    Wc = O.cast_weights(W, dtype)
    enc, _ = O.encoder(np.asarray(X, dtype), Wc, ocfg)
    trace = {}
    Y = O.beam_search(enc, Wc, ocfg, algo=algo, trace=trace)
    return Y, trace, enc


CASES = [
    dict(name='tiny_beam3_untied', cfg=dict(), lens=(60, 90, 75), seed=1, bias_scale=0.1, emb_scale=1.0),
    dict(name='tiny_greedy_untied', cfg=dict(beam_size=1), lens=(60, 90, 75, 30), seed=2, bias_scale=0.1, emb_scale=1.0),
    dict(name='tiny_beam10_tied', cfg=dict(beam_size=10, tie_embedding_and_softmax=True, max_target_length=25),
         lens=(100, 40), seed=3, bias_scale=0.2, emb_scale=1.0),
    dict(name='mid_beam4_relu', cfg=dict(hidden_units=256, num_heads=4, num_blocks=3, dst_vocab_size=300, beam_size=4,
                                         ff_activation='relu', max_target_length=30),
         lens=(150, 200, 120, 90, 201), seed=4, bias_scale=0.1, emb_scale=2.0),
    dict(name='short_T_limit', cfg=dict(beam_size=2, max_target_length=200), lens=(9, 6), seed=5, bias_scale=0.1, emb_scale=1.0),
    # beam > 16: the two-beam-group instantiations of the decode attention kernels
    dict(name='tiny_beam20', cfg=dict(beam_size=20, max_target_length=12), lens=(45, 70), seed=6, bias_scale=0.1, emb_scale=1.0),
    dict(name='tiny_beam32_tied', cfg=dict(beam_size=32, tie_embedding_and_softmax=True, max_target_length=8), lens=(33,), seed=7,
         bias_scale=0.2, emb_scale=1.0),
    # long decode: min(T + 50, max_target_length) + 1 = 151 steps over a ragged batch
    dict(name='tiny_long_decode', cfg=dict(beam_size=4, max_target_length=150), lens=(330, 120, 300), seed=8, bias_scale=0.1, emb_scale=1.0),
]


@pytest.mark.parametrize('case', CASES, ids=[c['name'] for c in CASES])
def test_encoder_and_beam_search_match_oracle(case):
    ocfg = tiny_config(**case['cfg'])
    W = O.random_weights(ocfg, seed=case['seed'], bias_scale=case['bias_scale'], emb_scale=case['emb_scale'])
    X = tiny_inputs(case['lens'], seed=case['seed'] + 10)
    eng = _engine(ocfg, W)
    try:
        # encoder
        enc_gpu, pad_gpu = eng.encode(X)
        enc64, pad64 = O.encoder(X.astype(np.float64), O.cast_weights(W, np.float64), ocfg)
        assert np.array_equal(pad_gpu.astype(bool), pad64)
        assert rel_err(enc_gpu, enc64) < 1e-4
        assert np.all(enc_gpu[pad64] == 0.0)                       # model.py:489
        # beam search: tokens identical to the float32 oracle (both decode algorithms) and the float64 one
        Y32, tr32, _ = _oracle_decode(X, W, ocfg, np.float32, 'kv')
        Y64, tr64, _ = _oracle_decode(X, W, ocfg, np.float64, 'ref')
        assert np.array_equal(Y32, Y64), 'oracle float32/float64 disagree: the case has a near-tie, pick another seed'
        steps = Y32.shape[1]
        Y, tr = eng.decode_trace(X, steps)
        assert Y.shape == Y32.shape
        assert np.array_equal(Y, Y32)
        for s in range(steps):
            assert np.array_equal(tr['tokens'][s], tr32['tokens'][s]), 'step %d tokens' % s
            assert np.array_equal(tr['parents'][s], tr32['parents'][s]), 'step %d parents' % s
            assert rel_err(tr['logits'][s], tr64['logits'][s]) < LOGIT_TOL
            assert rel_err_rows(tr['logits'][s], tr64['logits'][s]) < LOGIT_TOL      # per hypothesis row
            live = tr64['scores'][s] > -1e9
            assert np.allclose(tr['scores'][s][live], tr64['scores'][s][live], rtol=1e-4, atol=1e-4)
        # the plain entry point gives the same answer and is deterministic
        assert np.array_equal(eng.decode(X), Y)
        assert np.array_equal(eng.decode(X), Y)
    finally:
        eng.close()


def test_towers_match_reference_sharding():
    """num_gpus towers: dim-0 split of expand_feed_dict, per-shard T and padding with 3 (Q7, Q10)."""
    from asr_amd.model import Transformer
    ocfg = tiny_config(beam_size=2, max_target_length=15)
    W = O.random_weights(ocfg, seed=7, bias_scale=0.1)
    X = tiny_inputs((90, 60, 30, 45, 20), seed=8)
    for towers in (1, 2, 3, 7):
        m = Transformer(product_config(ocfg, towers), towers, devices=[0])
        m.build_test_model()
        m.restore(W)
        got = m.predict(X)
        want = O.build_test_model(X, W, ocfg, num_gpus=towers, algo='kv')
        assert np.array_equal(got, want), towers
        for e in m.engines:
            e.close()


def test_batch_row_independence_and_empty_batch():
    ocfg = tiny_config(beam_size=3)
    W = O.random_weights(ocfg, seed=9, bias_scale=0.1)
    X = tiny_inputs((75, 75, 75, 75, 75, 75), seed=11)
    eng = _engine(ocfg, W)
    try:
        full = eng.decode(X)
        sub = eng.decode(X[2:4])
        n = min(full.shape[1], sub.shape[1])
        assert np.array_equal(full[2:4, :n], sub[:, :n])
        assert np.all(full[2:4, n:] == 3) and np.all(sub[:, n:] == 3)
        assert eng.decode(np.zeros((0, 10, 320), np.float32)).shape == (0, 0)
    finally:
        eng.close()


@pytest.mark.parametrize('lens,beam,mtl', [((75, 40, 90, 33, 60, 81, 20), 3, 20), ((50, 64), 10, 12),
                                           ((30, 30, 30, 200, 200), 4, 60), ((45,) * 9, 2, 30)])
def test_two_half_batches_on_two_streams_give_the_same_hypotheses(lens, beam, mtl):
    """asr_config.sub_batches = 2: the beam search runs as two independent halves with their own loop state; a half
    that ends earlier is padded with 3 (the joint loop of model.py:305-312 ends when every utterance has)."""
    ocfg = tiny_config(beam_size=beam, max_target_length=mtl)
    W = O.random_weights(ocfg, seed=21, bias_scale=0.1)
    X = tiny_inputs(lens, seed=22)
    one, two = _engine(ocfg, W, sub_batches=1), _engine(ocfg, W, sub_batches=2)
    try:
        want, _, _ = _oracle_decode(X, W, ocfg, np.float32, 'kv')
        a = one.decode(X)
        b = two.decode(X)
        assert np.array_equal(a, want)
        assert b.shape == a.shape and np.array_equal(b, a)
        assert np.array_equal(two.decode(X), b)                        # and again: state of both streams is re-initialised
        assert np.array_equal(two.decode(X[:1]), a[:1, :two.decode(X[:1]).shape[1]])     # n = 1 falls back to one run
    finally:
        one.close()
        two.close()


def test_graph_replay_of_the_decode_step_gives_the_same_hypotheses(monkeypatch):
    """ASR_AMD_GRAPH=1 (the default from 320 beam rows up): steps 2.. of each sub-batch are captured once per call into a
    HIP graph and replayed; =0: every kernel launched by the host."""
    ocfg = tiny_config(beam_size=4, max_target_length=40)
    W = O.random_weights(ocfg, seed=41, bias_scale=0.1)
    X = tiny_inputs((75, 40, 90, 33, 60), seed=42)
    want, _, _ = _oracle_decode(X, W, ocfg, np.float32, 'kv')
    for parts, graph in ((1, '1'), (2, '1'), (3, '1'), (2, '0')):
        monkeypatch.setenv('ASR_AMD_GRAPH', graph)
        eng = _engine(ocfg, W, sub_batches=parts)
        try:
            assert np.array_equal(eng.decode(X), want), parts
            assert np.array_equal(eng.decode(X[1:4]), want[1:4, :eng.decode(X[1:4]).shape[1]]), parts      # another shape: captured anew
            assert np.array_equal(eng.decode(X), want), parts
        finally:
            eng.close()


@pytest.mark.parametrize('cls', ['glu', 'proj', 'qkv', 'ffn2', 'logits', 'self', 'cross', 'ln', 'topk', 'encoder'])
def test_launching_a_kernel_class_twice_changes_nothing(monkeypatch, cls):
    """ASR_AMD_DOUBLE (measurement aid, tools/marginal_cost.sh): every decode kernel of the class is launched twice; each is
    idempotent, so tokens, parents and logits of every step are bit-identical to a normal run."""
    ocfg = tiny_config(beam_size=4, max_target_length=12)
    W = O.random_weights(ocfg, seed=61, bias_scale=0.1)
    X = tiny_inputs((75, 40, 90), seed=62)
    monkeypatch.delenv('ASR_AMD_DOUBLE', raising=False)
    ref = _engine(ocfg, W, sub_batches=1)
    monkeypatch.setenv('ASR_AMD_DOUBLE', cls)
    twice = _engine(ocfg, W, sub_batches=1)
    try:
        Ya, ta = ref.decode_trace(X, 13)
        Yb, tb = twice.decode_trace(X, 13)
        assert np.array_equal(Ya, Yb)
        for s in range(len(ta['tokens'])):
            assert np.array_equal(ta['tokens'][s], tb['tokens'][s]) and np.array_equal(ta['parents'][s], tb['parents'][s])
            assert np.array_equal(ta['logits'][s], tb['logits'][s])
    finally:
        ref.close()
        twice.close()


def test_profile_modes_bracket_every_launch_and_leave_the_hypotheses_alone():
    """asr_set_profile 1 (every bracket waited for) and 2 (brackets left in the two queues): the same classes and launch
    counts, times > 0, the same hypotheses as an unprofiled call; self-attention's bytes are the live rows streamed."""
    ocfg = tiny_config(beam_size=4, max_target_length=14)
    W = O.random_weights(ocfg, seed=31, bias_scale=0.1)
    X = tiny_inputs((75, 40, 90, 33, 60, 81), seed=32)
    eng = _engine(ocfg, W, sub_batches=2)
    try:
        want = eng.decode(X)
        profs = {}
        for mode in (1, 2, 1, 2):
            eng.set_profile(mode)
            assert np.array_equal(eng.decode(X), want)
            profs[mode] = {p['name']: p for p in eng.get_profile()}
            eng.set_profile(0)
        assert set(profs[1]) == set(profs[2])
        for name, p in profs[2].items():
            assert p['launches'] == profs[1][name]['launches'], name
            assert p['flops'] == profs[1][name]['flops'] and p['bytes'] == profs[1][name]['bytes'], name
            if not name.startswith('('):
                assert 0.0 < p['ms'] < 1000.0 and 0.0 < profs[1][name]['ms'] < 1000.0, name
        steps = want.shape[1]
        sa = profs[2]['decoder_self_attention']
        assert sa['launches'] == 2 * steps * ocfg['num_blocks']
        # what it streams lies between 1 live row per history position (all beams share one history) and beam_size
        H, L, R = ocfg['hidden_units'], ocfg['num_blocks'], len(X) * ocfg['beam_size']
        fixed = lambda sq: steps * L * (4.0 * R * 3 * H * sq + 2.0 * R * H * 4 + 4.0 * R * H)     # slabs in, new K/V + context out
        hist = sum(len(X) * s for s in range(steps)) * L * 2.0 * H * 4
        assert sa['flops'] > 0
        assert fixed(1) + 0.5 * hist <= sa['bytes'] <= fixed(8) + hist * ocfg['beam_size']
        assert np.array_equal(eng.decode(X), want)
    finally:
        eng.close()


def test_device_front_end_matches_host_stacking():
    """asr_decode_dev with raw fbank == stacking on the host first (uniform lengths)."""
    import torch
    ocfg = tiny_config(beam_size=2, max_target_length=10)
    W = O.random_weights(ocfg, seed=12, bias_scale=0.1)
    raw = np.stack(O.synthetic_fbank(3, 100, seed=13))
    X = O.make_batch(list(raw))
    eng = _engine(ocfg, W)
    try:
        want = eng.decode(X)
        d_raw = torch.from_numpy(raw).cuda()
        torch.cuda.synchronize()
        got = eng.decode_dev(d_raw.data_ptr(), 3, 0, 0, raw=(100, 80, 3, 0, 3))
        assert np.array_equal(got, want)
    finally:
        eng.close()


def test_full_size_char_config_matches_oracle():
    """BASELINE config 2 architecture (H=1024, 16 heads, 6+6 blocks, V=3900, beam 10, tied) at a size the
    oracle finishes in seconds: 2 utterances x 3 s, 12 decode steps."""
    ocfg = O.make_config(beam_size=10, max_target_length=11)
    W = O.random_weights(ocfg, seed=21)
    X = O.make_batch(O.synthetic_fbank(2, [300, 240], seed=22))
    eng = _engine(ocfg, W)
    try:
        Y32, tr32, _ = _oracle_decode(X, W, ocfg, np.float32, 'kv')
        Y, tr = eng.decode_trace(X, Y32.shape[1])
        for s in range(Y32.shape[1]):
            assert rel_err(tr['logits'][s], tr32['logits'][s]) < LOGIT_TOL
            assert rel_err_rows(tr['logits'][s], tr32['logits'][s]) < LOGIT_TOL
        assert np.array_equal(Y, Y32)
    finally:
        eng.close()


FULL_SIZE_CONFIGS = [
    # BASELINE.json configs 3-5 at sizes the oracle finishes in seconds (full architecture, few utterances / steps)
    dict(name='pinyin_V1388_beam10', cfg=dict(dst_vocab_size=1388, beam_size=10, max_target_length=7), lens=[700, 230, 410], seed=31),
    dict(name='word_V28444_beam10', cfg=dict(dst_vocab_size=28444, beam_size=10, max_target_length=5), lens=[330, 250], seed=32),
    dict(name='yaml_beam13', cfg=dict(dst_vocab_size=3900, beam_size=13, max_target_length=5), lens=[200, 260], seed=34),
    dict(name='seame_V3205_30s', cfg=dict(dst_vocab_size=3205, beam_size=10, max_target_length=4), lens=[3000], seed=33),
]


@pytest.mark.parametrize('case', FULL_SIZE_CONFIGS, ids=[c['name'] for c in FULL_SIZE_CONFIGS])
def test_full_size_other_configs_match_oracle(case):
    """Pinyin / word / SEAME architectures of the reference's YAML files (hidden 1024, 16 heads, 6+6 blocks, GLU,
    tied): hypotheses identical to the float32 oracle, per-step logits within 1e-3 of it, ragged lengths."""
    ocfg = O.make_config(**case['cfg'])
    W = O.random_weights(ocfg, seed=case['seed'])
    X = O.make_batch(O.synthetic_fbank(len(case['lens']), case['lens'], seed=case['seed'] + 100))
    eng = _engine(ocfg, W)
    try:
        Y32, tr32, _ = _oracle_decode(X, W, ocfg, np.float32, 'kv')
        Y, tr = eng.decode_trace(X, Y32.shape[1])
        for s in range(Y32.shape[1]):
            assert rel_err(tr['logits'][s], tr32['logits'][s]) < LOGIT_TOL
            assert rel_err_rows(tr['logits'][s], tr32['logits'][s]) < LOGIT_TOL
            assert np.array_equal(tr['tokens'][s], tr32['tokens'][s]), 'step %d tokens' % s
            assert np.array_equal(tr['parents'][s], tr32['parents'][s]), 'step %d parents' % s
        assert np.array_equal(Y, Y32)
    finally:
        eng.close()


def test_full_size_batch64_properties():
    """BASELINE config 2 at full size (64 x 10 s, beam 10, max_target_length 200): properties that need no
    oracle run -- determinism, row independence, range, and the loop-length rule (Q7)."""
    ocfg = O.make_config(beam_size=10, max_target_length=200)
    from asr_amd import weights
    cfg = product_config(ocfg)
    W = weights.random_init(cfg, seed=0)
    from asr_amd.model import Engine
    eng = Engine(cfg, 0)
    eng.load_weights(W)
    try:
        X = O.make_batch(O.synthetic_fbank(64, 1000, seed=1))
        assert X.shape == (64, 334, 320)
        Y = eng.decode(X)
        assert Y.shape[0] == 64 and 1 <= Y.shape[1] <= 201
        assert Y.min() >= 0 and Y.max() < 3900
        Y2 = eng.decode(X)
        assert np.array_equal(Y, Y2)
        sub = eng.decode(X[5:7])
        n = min(sub.shape[1], Y.shape[1])
        assert np.array_equal(sub[:, :n], Y[5:7, :n])
    finally:
        eng.close()


def test_shared_gpu_handles_use_fewer_k_slices_and_give_the_same_hypotheses():
    """asr_config.shared_gpu = 1 (what EnginePool sets on its handles): from 448 beam rows up the decode-row GEMMs run with
    fewer K-slices (another summation order, same arithmetic).  Full-size char architecture, 64 short utterances x beam 10
    = 640 rows: tokens and parents of every step identical to the float32 oracle, logits within 1e-3, in both settings."""
    from asr_amd.model import Engine
    ocfg = O.make_config(beam_size=10, max_target_length=4)
    W = O.random_weights(ocfg, seed=51)
    X = O.make_batch(O.synthetic_fbank(64, [90 + 3 * (i % 7) for i in range(64)], seed=52))
    Y32, tr32, _ = _oracle_decode(X, W, ocfg, np.float32, 'kv')
    for shared in (0, 1):
        cfg = product_config(ocfg, sub_batches=1)
        cfg.test['shared_gpu'] = shared
        eng = Engine(cfg, 0)
        eng.load_weights(W)
        try:
            assert eng.cfg.shared_gpu == shared
            Y, tr = eng.decode_trace(X, Y32.shape[1])
            for s in range(Y32.shape[1]):
                assert rel_err(tr['logits'][s], tr32['logits'][s]) < LOGIT_TOL, (shared, s)
                assert np.array_equal(tr['tokens'][s], tr32['tokens'][s]), (shared, s)
                assert np.array_equal(tr['parents'][s], tr32['parents'][s]), (shared, s)
            assert np.array_equal(Y, Y32)
            assert np.array_equal(eng.decode(X), Y32)
        finally:
            eng.close()


def _targets(n, ty, vocab, seed):
    rng = np.random.default_rng(seed)
    Y = rng.integers(4, vocab, (n, ty)).astype(np.int32)
    for i in range(n):
        ln = int(rng.integers(1, ty + 1))
        Y[i, ln - 1] = 3                                            # </S> then PAD (0) as _create_target_batch makes them
        Y[i, ln:] = 0
    return Y


LOSS_CASES = [
    dict(name='tiny_untied', cfg=dict(), lens=(60, 90, 75), ty=7, seed=41),
    dict(name='tiny_tied_relu', cfg=dict(tie_embedding_and_softmax=True, ff_activation='relu'), lens=(100, 40), ty=1, seed=42),
    dict(name='mid_long_targets', cfg=dict(hidden_units=256, num_heads=4, num_blocks=3, dst_vocab_size=300), lens=(150, 200, 120, 90, 201),
         ty=150, seed=43),
]


@pytest.mark.parametrize('case', LOSS_CASES, ids=[c['name'] for c in LOSS_CASES])
def test_teacher_forced_loss_matches_oracle(case):
    """Evaluator.loss (evaluate.py:49-50): encoder + decoder_impl on shift_right(Y) + test_loss; float tolerance 1e-4
    relative on the summed loss (observed ~1e-6), towers add up (model.py:168-190)."""
    from asr_amd.model import Transformer
    ocfg = tiny_config(**case['cfg'])
    W = O.random_weights(ocfg, seed=case['seed'], bias_scale=0.1)
    X = tiny_inputs(case['lens'], seed=case['seed'] + 10)
    Y = _targets(len(case['lens']), case['ty'], ocfg.dst_vocab_size, case['seed'])
    want = O.loss_sum(X, Y, W, ocfg, dtype=np.float64)
    eng = _engine(ocfg, W)
    try:
        got = eng.loss(X, Y)
        assert abs(got - want) <= 1e-4 * abs(want), (got, want)
        assert eng.loss(X, Y) == got                                # deterministic
        assert eng.loss(X[:0], Y[:0]) == 0.0                        # empty shard: false_fn (model.py:184-187)
        assert eng.loss(X, np.zeros_like(Y)) == 0.0                 # all-PAD targets carry no loss
    finally:
        eng.close()
    for towers in (2, 3):
        m = Transformer(product_config(ocfg, towers), towers, devices=[0])
        m.build_test_model()
        m.restore(W)
        got_t = m.loss(X, Y)
        want_t = O.loss_sum(X, Y, W, ocfg, num_gpus=towers, dtype=np.float64)
        assert abs(got_t - want_t) <= 1e-4 * abs(want_t), (towers, got_t, want_t)
        for e in m.engines:
            e.close()


def test_teacher_forced_loss_full_size():
    """Char architecture (H=1024, 16 heads, 6+6 blocks, V=3900), 2 utterances, 40 target positions."""
    ocfg = O.make_config()
    W = O.random_weights(ocfg, seed=51)
    X = O.make_batch(O.synthetic_fbank(2, [300, 240], seed=52))
    Y = _targets(2, 40, 3900, 53)
    want = O.loss_sum(X, Y, W, ocfg, dtype=np.float32)
    eng = _engine(ocfg, W)
    try:
        got = eng.loss(X, Y)
        assert abs(got - want) <= 1e-4 * abs(want), (got, want)
    finally:
        eng.close()


def _upto_eos(row):
    row = list(row)
    return row[:row.index(3)] if 3 in row else row


@pytest.mark.parametrize('beam,mtl,lens_groups', [
    (3, 25, [(60, 62), (90,), (75, 75, 74)]),            # every group stops by its own EOS or the common limit
    (4, 60, [(15,), (100, 95), (30, 31)]),               # T + 50 < max_target_length for the short groups: their own limits bind
    (1, 30, [(40,), (41,), (80,)]),                      # greedy, one-utterance groups (config 1's shape)
    (1, 61, [(15,), (100,), (16, 14)]),                  # greedy: hypotheses that are still unfinished at their own limit
])
def test_joint_decode_of_groups_gives_each_group_its_own_hypotheses(beam, mtl, lens_groups):
    """asr_decode_groups: batches padded to different T decoded as ONE beam search.  Each group keeps its own T everywhere T
    is part of the arithmetic (input layer norm over (T, H), model.py:453; its keys; its loop limit min(T + 50,
    max_target_length), model.py:310), so every utterance's hypothesis - read up to the first 3, as indices_to_words reads it -
    must equal the oracle's for that group decoded on its own, and the engine's own group-by-group decode."""
    from asr_amd.model import Engine
    ocfg = tiny_config(beam_size=beam, max_target_length=mtl)
    W = O.random_weights(ocfg, seed=5, bias_scale=0.1)
    Xs = [tiny_inputs(lens, seed=20 + i) for i, lens in enumerate(lens_groups)]
    eng = Engine(product_config(ocfg), 0)
    eng.load_weights(W)
    try:
        got = eng.decode_groups(Xs)
        assert len(got) == len(Xs)
        for X, Yg in zip(Xs, got):
            want = O.build_test_model(X, W, ocfg, num_gpus=1)
            alone = eng.decode(X)
            assert Yg.shape[0] == X.shape[0]
            for u in range(X.shape[0]):
                assert _upto_eos(Yg[u]) == _upto_eos(want[u]), (X.shape, u)
                assert _upto_eos(Yg[u]) == _upto_eos(alone[u])
                assert len(_upto_eos(want[u])) <= min(X.shape[1] + 50, mtl) + 1
        if mtl == 60:
            # the case must exercise the per-utterance limit: the first group (T = 5) stops after min(5 + 50, 60) + 1 = 56 steps
            # with unfinished beams while the joint loop goes on for the others
            alone0 = eng.decode(Xs[0])
            assert alone0.shape[1] == 56 and got[0].shape[1] > 56
            assert np.array_equal(got[0][:, :56], alone0) and np.all(got[0][:, 56:] == 3)
        if mtl == 61:
            alone0 = eng.decode(Xs[0])                                 # T = 5: 56 tokens and no </S> among them
            assert alone0.shape[1] == 56 and 3 not in alone0[0] and got[0].shape[1] > 56
            assert np.array_equal(got[0][:, :56], alone0) and np.all(got[0][:, 56:] == 3)
        # a single group is the plain decode
        one = eng.decode_groups([Xs[1]])[0]
        assert np.array_equal(one, eng.decode(Xs[1]))
        assert np.array_equal(eng.decode(Xs[0]), O.build_test_model(Xs[0], W, ocfg, num_gpus=1))      # the handle is as before
    finally:
        eng.close()


def test_joint_decode_rejects_bad_group_descriptions_and_leaves_the_handle_usable():
    """asr_decode_groups through the C ABI: a group without utterances or frames, no groups, a null pointer, a wrong feature
    width and an output buffer that is too small are refused with an error code and a message; the handle still decodes."""
    import ctypes as C
    from asr_amd import _lib as L
    from asr_amd.model import Engine
    ocfg = tiny_config(beam_size=2, max_target_length=6)
    W = O.random_weights(ocfg, seed=2, bias_scale=0.1)
    eng = Engine(product_config(ocfg), 0)
    eng.load_weights(W)
    try:
        Xa, Xb = tiny_inputs((30, 33), seed=1), tiny_inputs((60,), seed=2)
        flat = np.concatenate([Xa.reshape(-1), Xb.reshape(-1)])
        Y = np.empty((3, 16), np.int32)
        ln = C.c_int32(0)

        def call(ng, tg, d=320, x=flat, cap=Y.size, groups=None):
            ng, tg = np.array(ng, np.int32), np.array(tg, np.int32)
            return L.lib.asr_decode_groups(eng._h, L.fptr(x) if x is not None else None, len(ng) if groups is None else groups,
                                           L.iptr(ng), L.iptr(tg), d, L.iptr(Y), cap, C.byref(ln))
        good = ([2, 1], [Xa.shape[1], Xb.shape[1]])
        assert call(*good) == 0 and 1 <= ln.value <= 7
        want = Y.reshape(-1)[:3 * ln.value].reshape(3, ln.value).copy()
        assert call([2, 0], good[1]) != 0 and b'group' in L.lib.asr_last_error(eng._h)
        assert call([2, 1], [Xa.shape[1], 0]) != 0
        assert call(*good, groups=0) != 0
        assert call(*good, x=None) != 0
        assert call(*good, d=319) != 0
        assert call(*good, cap=2) != 0 and b'Y too small' in L.lib.asr_last_error(eng._h)
        assert call(*good) == 0
        assert np.array_equal(Y.reshape(-1)[:3 * ln.value].reshape(3, ln.value), want)
        assert np.array_equal(eng.decode(Xb), O.build_test_model(Xb, W, ocfg, num_gpus=1))
    finally:
        eng.close()
