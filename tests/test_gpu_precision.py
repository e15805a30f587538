"""Arithmetic modes of the dense layers (include/asr_amd.h `asr_precision`; YAML extension key
`test.precision`).  The reference computes in float32 (model.py:35):

* `f32` (default)   exact-f32 matrix instructions.
* `f32_bf16x6`      float32 operands split exactly into three bf16 terms, six bf16 matrix products per k-step,
                    float32 accumulate.  Held to the SAME bars as `f32`: dense error vs float64 no larger than
                    the exact-f32 kernel's bar, hypotheses identical to the oracle.
* `bf16x3`          three products; tolerance 1e-4 on a dense layer.
* `bf16`            north_star's "bf16 MFMA" mode, tolerance stated separately: 2e-2 relative on a dense
                    layer, 5e-2 on first-step logits; hypotheses are NOT required to match (they need not on
                    random-init models, whose posteriors are nearly flat).
"""
import numpy as np
import pytest

from oracle import asr_oracle as O
from tests.helpers import product_config, rel_err, tiny_config, tiny_inputs

pytestmark = pytest.mark.gpu

DENSE_TOL = {'f32': 2e-6, 'f32_bf16x6': 2e-6, 'f32_f16x3': 2e-6, 'bf16x3': 1e-4, 'bf16': 2e-2}


def _engine(ocfg, precision, W=None):
    from asr_amd.model import Engine
    e = Engine(product_config(ocfg, precision=precision), 0)
    if W is not None:
        e.load_weights(W)
    return e


@pytest.fixture(scope='module', params=list(DENSE_TOL))
def mode_eng(request):
    e = _engine(tiny_config(), request.param)
    yield request.param, e
    e.close()


def _dense(eng, x, w, b, split_k=1):
    from asr_amd import _lib as L
    m, k = x.shape
    n = w.shape[0]
    y = np.empty((m, n), np.float32)
    L.check(eng._h, L.lib.asr_op_dense(eng._h, L.fptr(x), L.fptr(w), L.fptr(b), m, n, k, split_k, L.fptr(y)))
    return y


@pytest.mark.parametrize('m,n,k,split', [(37, 50, 128, 1), (640, 3072, 1024, 1), (640, 1024, 1024, 4),
                                         (640, 1024, 4096, 8), (30, 3900, 1024, 1), (2000, 1024, 320, 1),
                                         (129, 257, 64, 2), (2672, 1024, 1024, 1),
                                         # one k-tile; an odd k-tile count split unevenly (64 + 32); the 256-column tile
                                         (1, 64, 32, 1), (65, 130, 96, 2), (4100, 8200, 64, 1)])
def test_dense_modes(mode_eng, m, n, k, split):
    mode, eng = mode_eng
    rng = np.random.default_rng(m * 7 + n)
    x = rng.standard_normal((m, k)).astype(np.float32)
    w = (rng.standard_normal((n, k)) / np.sqrt(k)).astype(np.float32)
    b = rng.standard_normal(n).astype(np.float32)
    want = x.astype(np.float64) @ w.astype(np.float64).T + b
    got = _dense(eng, x, w, b, split)
    assert rel_err(got, want) < DENSE_TOL[mode]
    assert rel_err(_dense(eng, x, w, None, split), want - b) < DENSE_TOL[mode]


def test_split_modes_error_not_worse_than_exact_f32():
    """The split-product paths must be as accurate as the exact-f32 MFMA path (rms error vs float64)."""
    rng = np.random.default_rng(5)
    x = rng.standard_normal((640, 1024)).astype(np.float32)
    w = rng.uniform(-0.05, 0.05, (1024, 1024)).astype(np.float32)
    want = x.astype(np.float64) @ w.astype(np.float64).T
    rms = {}
    for mode in ('f32', 'f32_bf16x6', 'f32_f16x3'):
        e = _engine(tiny_config(), mode)
        try:
            rms[mode] = float(np.sqrt(np.mean((_dense(e, x, w, None).astype(np.float64) - want) ** 2)))
        finally:
            e.close()
    assert rms['f32_bf16x6'] <= 1.5 * rms['f32'], rms
    assert rms['f32_f16x3'] <= 1.5 * rms['f32'], rms


def test_f16x3_out_of_range_operands_are_redone_on_bf16x6():
    """fp16 holds magnitudes below 65504: an activation or weight beyond that must not change the result - the call is
    repeated on the bf16x6 kernels (asr_range_retries counts it)."""
    rng = np.random.default_rng(9)
    x = rng.standard_normal((70, 256)).astype(np.float32)
    w = (rng.standard_normal((96, 256)) / 16).astype(np.float32)
    e = _engine(tiny_config(), 'f32_f16x3')
    try:
        assert rel_err(_dense(e, x, w, None), x.astype(np.float64) @ w.astype(np.float64).T) < 2e-6
        assert e.range_retries() == 0
        xb = x.copy(); xb[3, 7] = 1.0e5; xb[69, 255] = -7.0e4          # activations outside fp16
        assert rel_err(_dense(e, xb, w, None), xb.astype(np.float64) @ w.astype(np.float64).T) < 2e-6
        assert e.range_retries() == 1
        wb = w.copy(); wb[95, 0] = 3.0e5                                 # a weight outside fp16
        assert rel_err(_dense(e, x, wb, None), x.astype(np.float64) @ wb.astype(np.float64).T) < 2e-6
        assert e.range_retries() == 2
        xs = x * 1.0e-6                                                  # rows so small that fp16 would lose relative accuracy
        assert rel_err(_dense(e, xs, w, None), xs.astype(np.float64) @ w.astype(np.float64).T) < 2e-6
        assert e.range_retries() == 3
        xr = np.maximum(x, 0.0); xr[xr < 0.05] *= 1.0e-3                 # sparse rows with tiny entries (ReLU-like): fine
        assert rel_err(_dense(e, xr, w, None), xr.astype(np.float64) @ w.astype(np.float64).T) < 2e-6
        assert e.range_retries() == 3
        xz = x.copy(); xz[5:9] = 0.0                                     # all-zero rows (padding) are exact, not "too small"
        assert rel_err(_dense(e, xz, w, None), xz.astype(np.float64) @ w.astype(np.float64).T) < 2e-6
        assert e.range_retries() == 3
    finally:
        e.close()


def test_f16x3_decode_with_out_of_range_features_matches_oracle():
    """Features scaled by 1e6 leave fp16's range in the first layer: the decode must still give the oracle's hypotheses."""
    ocfg = tiny_config()
    W = O.random_weights(ocfg, seed=2, bias_scale=0.1)
    X = tiny_inputs((60, 90, 75), seed=12) * np.float32(1.0e6)
    eng = _engine(ocfg, 'f32_f16x3', W)
    try:
        want = O.build_test_model(X, W, ocfg, num_gpus=1, algo='kv')
        Y = eng.decode(X)
        assert eng.range_retries() == 1
        assert Y.shape == want.shape and np.array_equal(Y, want)
    finally:
        eng.close()


def test_f16x3_cross_attention_keys_outside_fp16_range_are_redone_with_exact_products():
    """The default mode's decode-time cross-attention multiplies on the fp16 matrix pipe (three products per multiply, as its
    GEMMs do).  Keys of magnitude >= 65504 (the K rows of kv_transform scaled up; every GEMM operand stays in range) are
    caught where the K/V arrays are laid out: the call is repeated with the exact-f32 products and gives the oracle's
    hypotheses."""
    ocfg = tiny_config(beam_size=3, max_target_length=10)
    W = dict(O.random_weights(ocfg, seed=4, bias_scale=0.1))
    H = ocfg.hidden_units
    for l in range(ocfg.num_blocks):
        name = 'decoder/block_%d/decoder_vanilla_attention/kv_transform/kernel' % l
        w = W[name].copy()
        w[:H] *= np.float32(2.0e5)                                   # k = rows 0..H-1 (utils.py:618-619)
        assert np.abs(w).max() < 65504.0                           # the weights themselves still fit fp16
        W[name] = w
    X = tiny_inputs((60, 90, 75), seed=14)
    eng = _engine(ocfg, 'f32_f16x3', W)
    try:
        want = O.build_test_model(X, W, ocfg, num_gpus=1, algo='kv')
        want64 = O.build_test_model(X.astype(np.float64), O.cast_weights(W, np.float64), ocfg, num_gpus=1, algo='kv')
        assert np.array_equal(want, want64), 'near-tie in the oracle: pick another seed'
        Y = eng.decode(X)
        assert eng.range_retries() == 1
        assert Y.shape == want.shape and np.array_equal(Y, want)
        # and a batch that fits runs on the fp16 products without a repeat
        W2 = O.random_weights(ocfg, seed=4, bias_scale=0.1)
        eng2 = _engine(ocfg, 'f32_f16x3', W2)
        try:
            assert np.array_equal(eng2.decode(X), O.build_test_model(X, W2, ocfg, num_gpus=1, algo='kv'))
            assert eng2.range_retries() == 0
        finally:
            eng2.close()
    finally:
        eng.close()


def test_f16x3_self_attention_keys_outside_fp16_range_are_redone_with_exact_products():
    """The default mode's decode-time self-attention multiplies on the fp16 matrix pipe too (round 3).  Keys of magnitude >=
    65504 (the k rows of the decoder's qkv_transform scaled up; the weights themselves and every GEMM operand stay in range -
    k only enters the attention scores) are caught where the new position's q, k, v are assembled: the call is repeated with
    the exact-f32 products and gives the oracle's hypotheses."""
    ocfg = tiny_config(beam_size=3, max_target_length=10)
    W = dict(O.random_weights(ocfg, seed=6, bias_scale=0.1))
    H = ocfg.hidden_units
    for l in range(ocfg.num_blocks):
        name = 'decoder/block_%d/decoder_self_attention/qkv_transform/kernel' % l
        w = W[name].copy()
        w[H:2 * H] *= np.float32(2.0e5)                            # k = rows H..2H-1 (utils.py:612-615)
        assert np.abs(w).max() < 65504.0                           # the weights themselves still fit fp16
        W[name] = w
    X = tiny_inputs((60, 90, 75), seed=16)
    eng = _engine(ocfg, 'f32_f16x3', W)
    try:
        want = O.build_test_model(X, W, ocfg, num_gpus=1, algo='kv')
        want64 = O.build_test_model(X.astype(np.float64), O.cast_weights(W, np.float64), ocfg, num_gpus=1, algo='kv')
        assert np.array_equal(want, want64), 'near-tie in the oracle: pick another seed'
        Y = eng.decode(X)
        assert eng.range_retries() == 1
        assert Y.shape == want.shape and np.array_equal(Y, want)
    finally:
        eng.close()


def test_dense_glu_modes(mode_eng):
    from asr_amd import _lib as L
    mode, eng = mode_eng
    m, n, k = 640, 1024, 256
    rng = np.random.default_rng(n)
    x = rng.standard_normal((m, k)).astype(np.float32)
    w1 = (rng.standard_normal((n, k)) / np.sqrt(k)).astype(np.float32)
    w2 = (rng.standard_normal((n, k)) / np.sqrt(k)).astype(np.float32)
    b1 = rng.standard_normal(n).astype(np.float32)
    b2 = rng.standard_normal(n).astype(np.float32)
    y = np.empty((m, n), np.float32)
    L.check(eng._h, L.lib.asr_op_dense_glu(eng._h, L.fptr(x), L.fptr(w1), L.fptr(b1), L.fptr(w2), L.fptr(b2), m, n, k, L.fptr(y)))
    x64 = x.astype(np.float64)
    want = (x64 @ w1.astype(np.float64).T + b1) / (1.0 + np.exp(-(x64 @ w2.astype(np.float64).T + b2)))
    assert rel_err(y, want) < 10 * DENSE_TOL[mode]


CASES = [
    dict(name='tiny_beam3', cfg=dict(), lens=(60, 90, 75), seed=1),
    dict(name='tiny_beam10_tied', cfg=dict(beam_size=10, tie_embedding_and_softmax=True, max_target_length=25), lens=(100, 40), seed=3),
    dict(name='mid_beam4_relu', cfg=dict(hidden_units=256, num_heads=4, num_blocks=3, dst_vocab_size=300, beam_size=4,
                                         ff_activation='relu', max_target_length=30), lens=(150, 200, 120, 90, 201), seed=4),
]


@pytest.mark.parametrize('mode', ['f32_bf16x6', 'f32_f16x3'])
@pytest.mark.parametrize('case', CASES, ids=[c['name'] for c in CASES])
def test_split_modes_hypotheses_and_logits_match_oracle(case, mode):
    ocfg = tiny_config(**case['cfg'])
    W = O.random_weights(ocfg, seed=case['seed'], bias_scale=0.1)
    X = tiny_inputs(case['lens'], seed=case['seed'] + 10)
    eng = _engine(ocfg, mode, W)
    try:
        Wc = O.cast_weights(W, np.float64)
        enc64, _ = O.encoder(X.astype(np.float64), Wc, ocfg)
        enc_gpu, _ = eng.encode(X)
        assert rel_err(enc_gpu, enc64) < 1e-4
        trace = {}
        want = O.beam_search(enc64, Wc, ocfg, algo='kv', trace=trace)
        Y, tr = eng.decode_trace(X, 2)
        assert Y.shape == want.shape and np.array_equal(Y, want)
        assert rel_err(tr['logits'][0], trace['logits'][0]) < 1e-3      # north_star bar; observed ~1e-5
    finally:
        eng.close()


@pytest.mark.parametrize('mode', ['f32', 'f32_bf16x6', 'f32_f16x3'])
def test_nan_and_inf_features_do_not_poison_the_beam_search(mode):
    """Garbage in one utterance (NaN / inf features) must stay in that utterance: the call returns normally, every token
    is a vocabulary id, and the clean utterance of the same batch decodes as it does without the garbage (NaN logits
    once left the top-k without a candidate and the beam search indexed memory with stale values)."""
    ocfg = tiny_config(beam_size=3, max_target_length=12)
    W = O.random_weights(ocfg, seed=1, bias_scale=0.1)
    X = tiny_inputs((60, 60, 60), seed=2)
    want = O.build_test_model(X, W, ocfg, num_gpus=1, algo='kv')
    Xb = X.copy()
    Xb[1, 5, 7] = np.nan
    Xb[2, 9, 3] = np.inf
    eng = _engine(ocfg, mode, W)
    try:
        Y = eng.decode(Xb)
        assert Y.shape[0] == 3 and Y.min() >= 0 and Y.max() < ocfg.dst_vocab_size
        n = min(Y.shape[1], want.shape[1])
        assert np.array_equal(Y[0, :n], want[0, :n])
        assert np.array_equal(eng.decode(X), want)                # and the handle is still good
    finally:
        eng.close()


def test_bf16_mode_logits_within_stated_tolerance():
    ocfg = tiny_config()
    W = O.random_weights(ocfg, seed=1, bias_scale=0.1)
    X = tiny_inputs((60, 90, 75), seed=11)
    eng = _engine(ocfg, 'bf16', W)
    try:
        Wc = O.cast_weights(W, np.float64)
        enc64, _ = O.encoder(X.astype(np.float64), Wc, ocfg)
        enc_gpu, _ = eng.encode(X)
        assert rel_err(enc_gpu, enc64) < 5e-2
        trace = {}
        O.beam_search(enc64, Wc, ocfg, algo='kv', trace=trace)
        Y, tr = eng.decode_trace(X, 1)
        assert rel_err(tr['logits'][0], trace['logits'][0]) < 5e-2
        assert Y.dtype == np.int32 and Y.shape[0] == X.shape[0]
    finally:
        eng.close()


@pytest.mark.parametrize('beam,lens', [(3, (60, 90, 75)), (20, (45, 70))])
def test_bf16_mode_with_bf16_kv_caches_follows_the_oracle_over_several_steps(beam, lens):
    """`precision: bf16` also stores the self- and cross-attention K/V caches as bf16 (attention.hip KV16: half the bytes of
    the HBM-bound key loops; scores, soft-max and P.V stay float32).  Tolerance as for the mode: logits 5e-2 (max-abs /
    max-abs) against the float64 oracle for as long as the GPU's beam follows the oracle's; the history rows are read back
    from the bf16 cache from step 2 on, beam 20 runs the two-beam-group kernels.  Deterministic, and the two-stream split
    gives the same hypotheses."""
    ocfg = tiny_config(beam_size=beam, max_target_length=11)
    W = O.random_weights(ocfg, seed=3, bias_scale=0.1)
    X = tiny_inputs(lens, seed=13)
    from asr_amd.model import Engine
    eng = _engine(ocfg, 'bf16', W)
    two = Engine(product_config(ocfg, precision='bf16', sub_batches=2), 0)
    two.load_weights(W)
    try:
        Wc = O.cast_weights(W, np.float64)
        enc64, _ = O.encoder(X.astype(np.float64), Wc, ocfg)
        trace = {}
        want = O.beam_search(enc64, Wc, ocfg, algo='kv', trace=trace)
        Y, tr = eng.decode_trace(X, want.shape[1])
        followed = 0
        for s in range(want.shape[1]):
            assert rel_err(tr['logits'][s], trace['logits'][s]) < 5e-2, 'step %d' % s
            followed += 1
            if not (np.array_equal(tr['tokens'][s], trace['tokens'][s]) and np.array_equal(tr['parents'][s], trace['parents'][s])):
                break
        assert followed >= 2, 'the case must reach a step that reads history rows from the bf16 cache'
        assert np.array_equal(eng.decode(X), Y)
        assert np.array_equal(two.decode(X), Y)
        assert eng.device_bytes() > 0
    finally:
        eng.close()
        two.close()
