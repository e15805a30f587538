"""Known-answer tests that pin the CPU oracle.  The reference ships no tests or
fixtures and cannot run here (Python 2 + TensorFlow 1.x), so the oracle is "parity unpinned" by
the reference; what pins it is below: analytic values, float32 vs float64 agreement, and token
identity between the reference-shaped decode and the K/V-cache formulation.
"""
import json
import math
import os

import numpy as np

from oracle import asr_oracle as O
from tests.helpers import tiny_config, tiny_inputs

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')


def test_timing_signal_kat():
    s = O.timing_signal(8, 1024)
    assert s.shape == (8, 1024) and s.dtype == np.float32
    assert np.all(s[0, :512] == 0.0) and np.all(s[0, 512:] == 1.0)
    assert abs(float(s[1, 0]) - 0.8414709848) < 1e-7               # sin(1)
    assert abs(float(s[1, 512]) - math.cos(1.0)) < 1e-7
    inc = math.log(1e5) / 511
    assert abs(inc - 0.02253018682) < 1e-10
    assert abs(float(s[1, 511]) - math.sin(1e-5)) < 1e-11          # slowest timescale = 1e5 (Q2)
    s64 = O.timing_signal(300, 1024, dtype=np.float64)
    assert np.max(np.abs(O.timing_signal(300, 1024) - s64)) < 3e-5


def test_frame_stack_kat():
    x = np.arange(1, 13, dtype=np.float32).reshape(4, 3)           # the 4x3 matrix of fea.py:166
    sp = O.tf_fea_splice(x, 1, 1)
    assert sp.shape == (4, 9)
    assert np.array_equal(sp[0], [0, 0, 0, 1, 2, 3, 4, 5, 6])
    assert np.array_equal(sp[3], [7, 8, 9, 10, 11, 12, 0, 0, 0])
    st = O.frame_stack(x, left=3, right=0, sub=3)
    assert st.shape == (2, 12)
    assert np.array_equal(st[0], [0] * 9 + [1, 2, 3])
    assert np.array_equal(st[1], [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12])
    assert O.frame_stack(np.ones((1000, 80), np.float32)).shape == (334, 320)


def test_layer_norm_kats():
    bias = np.linspace(-1, 1, 16).astype(np.float32)
    scale = np.linspace(0.5, 2, 16).astype(np.float32)
    y = O.layer_norm(np.full((3, 16), 7.5, np.float32), scale, bias)
    assert np.array_equal(y, np.broadcast_to(bias, (3, 16)))       # constant row -> bias
    x = np.array([[1.0, -1.0, 1.0, -1.0]], np.float32)
    y = O.layer_norm(x, np.ones(4, np.float32), np.zeros(4, np.float32))
    assert np.allclose(y, x / math.sqrt(1 + 1e-6), atol=1e-7)      # biased variance, eps 1e-6
    # contrib LN normalises over (T, H) jointly (Q1): per-utterance mean 0 / var 1 over all elements
    rng = np.random.default_rng(0)
    z = rng.standard_normal((2, 5, 8)) * 3 + 2
    out = O.contrib_layer_norm(z, np.ones(8), np.zeros(8))
    assert np.allclose(out.reshape(2, -1).mean(1), 0, atol=1e-12)
    assert np.allclose(out.reshape(2, -1).var(1), 1, atol=1e-9)
    assert not np.allclose(out.var(-1), 1, atol=1e-2)


def test_masked_softmax_exact_zeros():
    q = np.ones((1, 1, 2, 4), np.float32)
    k = np.random.default_rng(1).standard_normal((1, 1, 3, 4)).astype(np.float32)
    v = np.eye(3, 4, dtype=np.float32)[None, None]
    bias = np.array([0, -1e9, 0], np.float32)[None, None, None, :]
    w = O.softmax(q @ k.transpose(0, 1, 3, 2) + bias)
    assert np.all(w[..., 1] == 0.0)                                # -1e9 in float32 -> exactly zero weight
    out = O.dot_product_attention(q, k, v, bias)
    assert np.all(out[..., 1] == 0.0)


def test_expand_shards():
    assert O.expand_shards(10, 4) == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert O.expand_shards(2, 4) == [(0, 1), (1, 2), (2, 2), (2, 2)]
    assert O.expand_shards(8, 1) == [(0, 8)]


def test_top_k_order():
    z = np.array([[1.0, 3.0, 3.0, 2.0, 3.0]], np.float32)
    v, i = O.top_k(z, 3)
    assert i.tolist() == [[1, 2, 4]] and v.tolist() == [[3.0, 3.0, 3.0]]


def test_beam_bookkeeping_hand_table():
    K = 2
    scores = np.array([-1.0, -2.0], np.float32)
    fin = np.array([False, True])
    cs = np.array([[-0.5, -0.7], [-0.1, -0.2]], np.float32)
    ct = np.array([[5, 6], [7, 8]], np.int32)
    ns, tok, par, nf = O.beam_bookkeeping(scores, fin, cs, ct, K)
    # finished beam 1: candidates (3, 0) and (3, -1e10); totals: [-1.5, -1.7, -2.0, -1e10]
    assert np.array_equal(ns, np.float32([-1.0, -1.0]) + np.float32([-0.5, -0.7]))
    assert tok.tolist() == [5, 6] and par.tolist() == [0, 0] and nf.tolist() == [False, False]
    # -1e10 + logp == -1e10 in float32 for |logp| < 512 (Q6)
    assert np.float32(-1e10) + np.float32(-300.0) == np.float32(-1e10)
    # tie between two candidates -> lower flat index wins
    scores = np.array([0.0, 0.0], np.float32)
    cs = np.array([[-1.0, -3.0], [-1.0, -2.0]], np.float32)
    ns, tok, par, nf = O.beam_bookkeeping(scores, np.array([False, False]), cs, ct, K)
    assert par.tolist() == [0, 1] and tok.tolist() == [5, 7]
    # a finished hypothesis survives with its score and emits </S>
    ns, tok, par, nf = O.beam_bookkeeping(np.array([-0.2, -5.0], np.float32), np.array([True, False]), cs, ct, K)
    assert tok[0] == 3 and par[0] == 0 and ns[0] == np.float32(-0.2) and nf[0]


def test_finalize_length_penalty_counts_bos():
    B, K = 1, 2
    preds = np.array([[2, 5, 3, 3], [2, 6, 7, 8]], np.int32)       # lengths incl <S>: 2 and 4
    scores = np.array([-2.0, -2.5], np.float32)
    lp = [((5 + 2) / 6) ** 0.6, ((5 + 4) / 6) ** 0.6]
    want = int(np.argmax([-2.0 / lp[0], -2.5 / lp[1]]))
    Y = O.finalize(preds, scores, B, K, 0.6)
    assert np.array_equal(Y[0], preds[want][1:])
    assert want == 0
    Y = O.finalize(preds, np.array([-2.0, -2.3], np.float32), B, K, 0.6)
    assert np.array_equal(Y[0], preds[1][1:])                      # the longer hypothesis wins after the penalty


def test_variable_table_matches_survey_a1():
    s = O.variable_shapes(O.make_config())
    assert s['encoder/src_change/kernel'] == (1024, 320)
    assert s['encoder/block_5/encoder_self_attention/qkv_transform/kernel'] == (3072, 1024)
    assert s['encoder/block_0/ff_hidden/dense/arg2/kernel'] == (4096, 1024)
    assert s['decoder/block_3/decoder_vanilla_attention/kv_transform/bias'] == (2048,)
    assert s['decoder/block_0/layer_norm_2/layer_norm_scale'] == (1024,)
    assert s['decoder/dst_embedding/kernel'] == (3900, 1024)
    enc = sum(int(np.prod(v)) for k, v in s.items() if k.startswith('encoder'))
    dec = sum(int(np.prod(v)) for k, v in s.items() if k.startswith('decoder'))
    assert abs(enc / 1e6 - 101.0) < 0.5 and abs((dec - 3900 * 1024) / 1e6 - 125.8) < 0.5


def test_reference_shaped_and_kv_decode_agree_and_f32_f64_agree():
    for cfgkw, lens, seed in [(dict(), (60, 90, 75), 1), (dict(beam_size=1), (45, 33), 2),
                              (dict(beam_size=4, tie_embedding_and_softmax=True, max_target_length=12), (50, 70), 3)]:
        ocfg = tiny_config(**cfgkw)
        W = O.random_weights(ocfg, seed=seed, bias_scale=0.1)
        X = tiny_inputs(lens, seed=seed + 10)
        ys = [O.build_test_model(X, W, ocfg, 1, algo, dt) for algo in ('ref', 'kv') for dt in (np.float32, np.float64)]
        for y in ys[1:]:
            assert np.array_equal(ys[0], y)
        assert ys[0].shape[1] <= min(X.shape[1] + 50, ocfg.max_target_length) + 1            # Q7


def test_towers_change_shard_lengths():
    ocfg = tiny_config(beam_size=2, max_target_length=12)
    W = O.random_weights(ocfg, seed=7, bias_scale=0.1)
    X = tiny_inputs((90, 60, 30, 45, 20), seed=8)
    y1 = O.build_test_model(X, W, ocfg, 1)
    y3 = O.build_test_model(X, W, ocfg, 3)
    assert y1.shape[0] == y3.shape[0] == 5
    assert O.build_test_model(X[:0], W, ocfg, 1).shape == (0, 0)


def test_golden_fixture():
    """tests/golden/tiny_decode.json was produced by tests/golden/make_golden.py from this oracle; it
    freezes the oracle's behaviour so that later edits cannot drift silently (it is NOT reference output)."""
    with open(os.path.join(GOLDEN, 'tiny_decode.json')) as f:
        g = json.load(f)
    for case in g['cases']:
        ocfg = tiny_config(**case['cfg'])
        W = O.random_weights(ocfg, seed=case['seed'], bias_scale=case['bias_scale'], emb_scale=case['emb_scale'])
        X = tiny_inputs(tuple(case['lens']), seed=case['seed'] + 10)
        Y = O.build_test_model(X, W, ocfg, case['towers'], 'kv')
        assert Y.tolist() == case['Y'], case['name']
        enc, _ = O.encoder(X, W, ocfg)
        assert abs(float(np.abs(enc).sum()) - case['enc_abs_sum']) < 1e-3 * case['enc_abs_sum']


# ---- feature front end (third_party/feat_convert/io/fea.py) ---------------------------------------------------------
def test_fea_delta_known_answers():
    """A ramp x[t] = a*t has delta a and delta-delta 0 away from the edges (window weights: 2*0.1 + 4*0.2 = 1;
    -0.1 - 0.08 + 0.02 + 0.16 = 0); a constant has both zero everywhere (edge frames are repeated)."""
    t = np.arange(20, dtype=np.float64)[:, None] * np.array([[1.0, -2.5]])
    out = O.fea_add_delt(t)
    assert out.shape == (20, 6)
    np.testing.assert_allclose(out[2:-2, 2:4], np.broadcast_to([1.0, -2.5], (16, 2)), atol=1e-12)
    np.testing.assert_allclose(out[4:-4, 4:6], 0.0, atol=1e-12)
    const = np.full((5, 3), 7.0)
    np.testing.assert_allclose(O.fea_add_delt(const)[:, 3:], 0.0, atol=1e-12)
    # first frame of the ramp: l1 = l2 = x[0] -> (x1 - x0)*0.1 + (x2 - x0)*0.2 = 0.5a
    np.testing.assert_allclose(out[0, 2:4], [0.5, -1.25], atol=1e-12)


def test_fea_delta_equals_written_out_padding():
    """The clamped-index form used by the oracle == the repeated one-frame symmetric padding written out,
    bit for bit in float32 (also for utterances shorter than the window)."""
    for n in (1, 2, 3, 9, 50):
        x = np.random.default_rng(n).standard_normal((n, 4)).astype(np.float32)
        p = x
        for _ in range(2):
            p = np.pad(p, [[1, 1], [0, 0]], 'symmetric')
        d1 = (p[3:n + 3] - p[1:n + 1]) * 0.1 + (p[4:n + 4] - p[:n]) * 0.2
        assert d1.dtype == np.float32 and np.array_equal(d1, O.fea_delt1(x))
        for _ in range(2):
            p = np.pad(p, [[1, 1], [0, 0]], 'symmetric')
        w = [p[i:n + i] for i in range(9)]
        d2 = -0.1 * w[4] - 0.04 * (w[3] + w[5]) + 0.01 * (w[2] + w[6]) + 0.04 * (w[1] + w[0] + w[8] + w[7])
        assert np.array_equal(d2, O.fea_delt2(x))


def test_fea_cmvn_and_splice_known_answers():
    x = np.random.default_rng(5).standard_normal((200, 6)) * 3 + 2
    y = O.fea_cmvn_utt(x)
    np.testing.assert_allclose(y.mean(axis=0), 0, atol=1e-12)
    np.testing.assert_allclose(y.var(axis=0), 1, atol=1e-9)
    np.testing.assert_allclose(O.fea_cmvn_global(x, np.full(6, 2.0), np.full(6, 4.0)), (x - 2) / 4)
    z = np.arange(12, dtype=np.float32).reshape(6, 2)
    assert np.array_equal(O.splice_general(z, 3, 0), O.tf_fea_splice(z, 3, 0))
    assert np.array_equal(O.splice_general(z, 3, 0, sub=3), O.frame_stack(z, 3, 0, 3))
    e = O.splice_general(z, 1, 1, edge=True)
    assert np.array_equal(e[0], [0, 1, 0, 1, 2, 3]) and np.array_equal(e[-1], [8, 9, 10, 11, 10, 11])
    assert np.array_equal(O.splice_general(z, 1, 0, sub=2, offset=1)[:, 2:], z[1::2])
