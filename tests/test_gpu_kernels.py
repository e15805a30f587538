"""Per-kernel parity: each HIP kernel, called through the C ABI probes, against the CPU oracle.

Bars: bit-exact for integer / index outputs (top-k indices, beam parents/tokens, frame stacking is a
pure copy so also bit-exact); float32 outputs within the stated relative tolerance of the float64
oracle (north_star: 1e-3 on logits; the kernels are held to 2e-5 or tighter).
"""
import ctypes as C

import numpy as np
import pytest

from oracle import asr_oracle as O
from tests.helpers import product_config, rel_err, tiny_config

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def eng():
    from asr_amd.model import Engine
    e = Engine(product_config(tiny_config()), 0)
    yield e
    e.close()


@pytest.fixture(scope='module')
def L():
    from asr_amd import _lib
    return _lib


def _dense(L, eng, x, w, b, split_k=1):
    m, k = x.shape
    n = w.shape[0]
    y = np.empty((m, n), np.float32)
    L.check(eng._h, L.lib.asr_op_dense(eng._h, L.fptr(x), L.fptr(w), L.fptr(b), m, n, k, split_k, L.fptr(y)))
    return y


@pytest.mark.parametrize('left,right,sub,n', [(3, 0, 3, 1000), (3, 0, 3, 301), (1, 1, 1, 17), (0, 0, 2, 9), (3, 0, 3, 2)])
def test_frame_stack_bit_exact(L, eng, left, right, sub, n):
    rng = np.random.default_rng(n)
    raw = rng.standard_normal((n, 80)).astype(np.float32)
    got = eng.stack_frames(raw, left, right, sub)
    want = O.frame_stack(raw, left, right, sub)
    assert got.shape == want.shape
    assert np.array_equal(got, want)


# ---- feature front end (third_party/feat_convert/io/fea.py) ---------------------------------------------------------
FEA_SHAPES = [(1000, 80), (3000, 80), (301, 40), (9, 13), (4, 80), (1, 80), (2, 3)]


@pytest.mark.parametrize('n,dim', FEA_SHAPES)
def test_fea_add_delt_bit_exact(eng, n, dim):
    """Delta / delta-delta windows of fea.py:20-52: same float32 operations in the same order -> identical bits."""
    raw = np.random.default_rng(n * 131 + dim).standard_normal((n, dim)).astype(np.float32)
    got = eng.fea_add_delt(raw)
    want = O.fea_add_delt(raw)
    assert got.shape == want.shape == (n, 3 * dim) and want.dtype == np.float32
    assert np.array_equal(got, want)


@pytest.mark.parametrize('n,dim', FEA_SHAPES)
def test_fea_cmvn_global_bit_exact(eng, n, dim):
    rng = np.random.default_rng(n * 7 + dim)
    raw = (rng.standard_normal((n, dim)) * 3 + 1).astype(np.float32)
    mean = rng.standard_normal(dim).astype(np.float32)
    var = (rng.random(dim) + 0.5).astype(np.float32)
    got = eng.fea_cmvn_global(raw, mean, var)
    assert np.array_equal(got, O.fea_cmvn_global(raw, mean, var))


@pytest.mark.parametrize('n,dim', FEA_SHAPES[:5])
def test_fea_cmvn_utt(eng, n, dim):
    """Floating-point reduction: within 2e-5 (relative, max-norm) of the float64 evaluation of fea.py:134-140."""
    rng = np.random.default_rng(n + dim)
    raw = (rng.standard_normal((n, dim)) * rng.uniform(0.5, 4, dim) + rng.uniform(-2, 2, dim)).astype(np.float32)
    got = eng.fea_cmvn_utt(raw)
    want = O.fea_cmvn_utt(raw.astype(np.float64))
    assert rel_err(got, want) < 2e-5
    assert abs(got.mean(axis=0)).max() < 1e-4 and abs(got.std(axis=0) - 1).max() < 1e-3


@pytest.mark.parametrize('left,right,sub,offset,edge,n', [(3, 0, 3, 0, False, 1000), (3, 0, 3, 1, False, 1000),
                                                           (3, 0, 3, 2, True, 301), (5, 5, 1, 0, True, 40),
                                                           (2, 4, 2, 1, False, 7), (3, 3, 1, 0, True, 1),
                                                           (0, 0, 4, 3, False, 4)])
def test_fea_splice_bit_exact(eng, left, right, sub, offset, edge, n):
    raw = np.random.default_rng(n).standard_normal((n, 80)).astype(np.float32)
    got = eng.fea_splice(raw, left, right, sub=sub, offset=offset, edge=edge)
    want = O.splice_general(raw, left, right, sub, offset, edge)
    assert got.shape == want.shape
    assert np.array_equal(got, want)
    if not edge and offset == 0:
        assert np.array_equal(got, eng.stack_frames(raw, left, right, sub))


def test_fea_bad_arguments(eng):
    with pytest.raises(ValueError):
        eng.fea_add_delt(np.zeros((0, 80), np.float32))
    with pytest.raises(ValueError):
        eng.fea_cmvn_global(np.zeros((3, 80), np.float32), np.zeros(79), np.ones(80))
    with pytest.raises(ValueError):
        eng.fea_splice(np.zeros((3, 80), np.float32), 1, 1, offset=3)
    with pytest.raises(RuntimeError):
        eng.fea_splice(np.zeros((3, 80), np.float32), -1, 1)


@pytest.mark.parametrize('m,n,k,split', [(37, 50, 128, 1), (640, 3072, 1024, 1), (640, 1024, 1024, 4),
                                         (640, 1024, 4096, 8), (30, 3900, 1024, 1), (2000, 1024, 320, 1),
                                         (1, 64, 4, 1), (129, 257, 36, 2), (21376 // 8, 1024, 1024, 1)])
def test_dense(L, eng, m, n, k, split):
    rng = np.random.default_rng(m * 7 + n)
    x = rng.standard_normal((m, k)).astype(np.float32)
    w = (rng.standard_normal((n, k)) / np.sqrt(k)).astype(np.float32)
    b = rng.standard_normal(n).astype(np.float32)
    got = _dense(L, eng, x, w, b, split)
    want = x.astype(np.float64) @ w.astype(np.float64).T + b
    assert rel_err(got, want) < 2e-6
    got_nb = _dense(L, eng, x, w, None, split)
    assert rel_err(got_nb, want - b) < 2e-6


def test_dense_is_row_position_independent(L, eng):
    """A row's result must not depend on where it sits in the batch (decode(X[a:b]) == decode(X)[a:b])."""
    rng = np.random.default_rng(0)
    x = rng.standard_normal((300, 256)).astype(np.float32)
    w = rng.standard_normal((192, 256)).astype(np.float32)
    full = _dense(L, eng, x, w, None)
    part = _dense(L, eng, x[130:177], w, None)
    assert np.array_equal(full[130:177], part)


@pytest.mark.parametrize('m,n,k', [(640, 4096, 1024), (45, 512, 128), (3000, 512, 128)])
def test_dense_glu(L, eng, m, n, k):
    rng = np.random.default_rng(n)
    x = rng.standard_normal((m, k)).astype(np.float32)
    w1 = (rng.standard_normal((n, k)) / np.sqrt(k)).astype(np.float32)
    w2 = (rng.standard_normal((n, k)) / np.sqrt(k)).astype(np.float32)
    b1 = rng.standard_normal(n).astype(np.float32)
    b2 = rng.standard_normal(n).astype(np.float32)
    y = np.empty((m, n), np.float32)
    L.check(eng._h, L.lib.asr_op_dense_glu(eng._h, L.fptr(x), L.fptr(w1), L.fptr(b1), L.fptr(w2), L.fptr(b2), m, n, k, L.fptr(y)))
    x64 = x.astype(np.float64)
    want = (x64 @ w1.T.astype(np.float64) + b1) * (1.0 / (1.0 + np.exp(-(x64 @ w2.T.astype(np.float64) + b2))))
    assert rel_err(y, want) < 5e-6


@pytest.mark.parametrize('rows,hidden', [(7, 128), (640, 1024), (33, 2048), (5, 64), (100, 320)])
def test_add_layer_norm(L, eng, rows, hidden):
    rng = np.random.default_rng(rows)
    x = rng.standard_normal((rows, hidden)).astype(np.float32)
    d = rng.standard_normal((rows, hidden)).astype(np.float32)
    db = rng.standard_normal(hidden).astype(np.float32)
    sc = (1 + 0.3 * rng.standard_normal(hidden)).astype(np.float32)
    bi = rng.standard_normal(hidden).astype(np.float32)
    y = np.empty_like(x)
    L.check(eng._h, L.lib.asr_op_add_layer_norm(eng._h, L.fptr(x), L.fptr(d), L.fptr(db), L.fptr(sc), L.fptr(bi), rows, hidden, L.fptr(y)))
    want = O.layer_norm((x.astype(np.float64) + (d.astype(np.float64) + db)), sc.astype(np.float64), bi.astype(np.float64))
    assert rel_err(y, want) < 2e-6
    # KAT: a constant row normalises to exactly the bias
    xc = np.full((rows, hidden), 3.25, np.float32)
    z = np.zeros_like(xc)
    L.check(eng._h, L.lib.asr_op_add_layer_norm(eng._h, L.fptr(xc), L.fptr(z), None, L.fptr(sc), L.fptr(bi), rows, hidden, L.fptr(y)))
    assert np.array_equal(y, np.broadcast_to(bi, y.shape))


@pytest.mark.parametrize('n,t', [(3, 30), (2, 334), (1, 1), (5, 77)])
def test_input_norm_timing(L, eng, n, t):
    hidden = eng.hidden
    rng = np.random.default_rng(t)
    x = (rng.standard_normal((n, t, hidden)) * 2 + 0.7).astype(np.float32)
    g = (1 + 0.3 * rng.standard_normal(hidden)).astype(np.float32)
    b = rng.standard_normal(hidden).astype(np.float32)
    y = np.empty_like(x)
    L.check(eng._h, L.lib.asr_op_input_norm_timing(eng._h, L.fptr(x), L.fptr(g), L.fptr(b), n, t, hidden, L.fptr(y)))
    want = O.contrib_layer_norm(x.astype(np.float64), g.astype(np.float64), b.astype(np.float64)) + \
        O.timing_signal(t, hidden, dtype=np.float32).astype(np.float64)[None]
    assert rel_err(y, want) < 5e-6
    # the timing table itself: zero input with gamma=0, beta=0 returns exactly the signal
    if t > 1:
        x0 = rng.standard_normal((n, t, hidden)).astype(np.float32)
        zero = np.zeros(hidden, np.float32)
        L.check(eng._h, L.lib.asr_op_input_norm_timing(eng._h, L.fptr(x0), L.fptr(zero), L.fptr(zero), n, t, hidden, L.fptr(y)))
        assert np.array_equal(y[0], O.timing_signal(t, hidden, dtype=np.float32))


@pytest.mark.parametrize('n,t,lens', [(3, 30, (20, 30, 25)), (2, 334, (334, 200)), (1, 129, (129,)), (2, 64, (64, 1)),
                                      (1, 5, (5,))])
def test_encoder_attention(L, eng, n, t, lens):
    hidden, heads = eng.hidden, eng.cfg.num_heads
    rng = np.random.default_rng(t)
    qkv = rng.standard_normal((n, t, 3 * hidden)).astype(np.float32)
    pad = np.zeros((n, t), np.int32)
    for i, l in enumerate(lens):
        pad[i, l:] = 1
    ctx = np.empty((n, t, hidden), np.float32)
    L.check(eng._h, L.lib.asr_op_encoder_attention(eng._h, L.fptr(qkv), L.iptr(pad), n, t, L.fptr(ctx)))
    q, k, v = np.split(qkv.astype(np.float64), 3, axis=2)
    q = O.split_heads(q, heads) * 64 ** -0.5
    bias = (pad.astype(np.float64) * -1e9)[:, None, None, :]
    want = O.combine_heads(O.dot_product_attention(q, O.split_heads(k, heads), O.split_heads(v, heads), bias))
    assert rel_err(ctx, want) < 5e-6


def test_encoder_attention_per_precision_mode_and_range_guard(L):
    """The many-query attention runs on the arithmetic of the handle's mode (attention.hip launch_mha): `f32_f16x3` = three
    fp16 matrix products per multiply (bar: the exact-f32 kernel's own, 5e-6 of the float64 evaluation; observed 5e-7 against
    6e-7 .. 1.1e-6 for the exact-f32 instruction), `f32` / `f32_bf16x6` = the exact-f32 instruction, `bf16` = K and V rounded
    to bf16 (2e-2).  In `f32_f16x3` an operand outside fp16's range (|x| >= 65504) sets the range flag and the call is
    repeated on the exact-f32 instruction: same bar, one retry counted."""
    from asr_amd.model import Engine
    from tests.helpers import product_config
    ocfg = O.make_config(beam_size=2, max_target_length=4)
    rng = np.random.default_rng(77)
    n, t, lens = 3, 150, (150, 97, 5)
    for prec, bar in (('f32_f16x3', 5e-6), ('f32', 5e-6), ('f32_bf16x6', 5e-6), ('bf16', 2e-2)):
        e = Engine(product_config(ocfg, precision=prec), 0)
        e.load_weights(O.random_weights(ocfg, seed=1))
        try:
            hidden, heads = e.hidden, e.cfg.num_heads
            for scale_k in ((1.0, 3.0e5) if prec == 'f32_f16x3' else (1.0,)):
                qkv = rng.standard_normal((n, t, 3 * hidden)).astype(np.float32)
                qkv[:, :, :hidden] *= np.float32(1.0 / scale_k)                 # (keeps the logits in a sane range)
                qkv[:, :, hidden:2 * hidden] *= np.float32(scale_k)             # k up to ~1.2e6: outside fp16
                pad = np.zeros((n, t), np.int32)
                for i, l in enumerate(lens):
                    pad[i, l:] = 1
                ctx = np.empty((n, t, hidden), np.float32)
                before = e.range_retries()
                L.check(e._h, L.lib.asr_op_encoder_attention(e._h, L.fptr(qkv), L.iptr(pad), n, t, L.fptr(ctx)))
                assert e.range_retries() - before == (1 if scale_k > 1.0 else 0), (prec, scale_k)
                q, k, v = np.split(qkv.astype(np.float64), 3, axis=2)
                q = O.split_heads(q, heads) * 64 ** -0.5
                bias = (pad.astype(np.float64) * -1e9)[:, None, None, :]
                want = O.combine_heads(O.dot_product_attention(q, O.split_heads(k, heads), O.split_heads(v, heads), bias))
                assert rel_err(ctx, want) < bar, (prec, scale_k, rel_err(ctx, want))
        finally:
            e.close()


def test_encoder_attention_mask_gives_exact_zero_weight(L, eng):
    """Padded keys must not leak: changing their K/V rows leaves the context bit-identical."""
    hidden = eng.hidden
    n, t = 1, 70
    rng = np.random.default_rng(5)
    qkv = rng.standard_normal((n, t, 3 * hidden)).astype(np.float32)
    pad = np.zeros((n, t), np.int32)
    pad[0, 40:] = 1
    a = np.empty((n, t, hidden), np.float32)
    b = np.empty_like(a)
    L.check(eng._h, L.lib.asr_op_encoder_attention(eng._h, L.fptr(qkv), L.iptr(pad), n, t, L.fptr(a)))
    qkv2 = qkv.copy()
    qkv2[0, 40:, hidden:] = rng.standard_normal((30, 2 * hidden)).astype(np.float32)
    L.check(eng._h, L.lib.asr_op_encoder_attention(eng._h, L.fptr(qkv2), L.iptr(pad), n, t, L.fptr(b)))
    assert np.array_equal(a, b)


@pytest.mark.parametrize('rows,vocab,k', [(640, 3900, 10), (7, 50, 3), (64, 28444, 10), (5, 1388, 13), (3, 10, 10), (4, 4, 1),
                                          (9, 2048, 10), (9, 4096, 32), (3, 2052, 5), (3, 8, 8), (9, 3205, 10), (5, 4100, 10)])
def test_log_softmax_topk(L, eng, rows, vocab, k):
    rng = np.random.default_rng(vocab)
    x = (rng.standard_normal((rows, vocab)) * 3).astype(np.float32)
    x[0, :] = np.round(x[0, :])                       # many exact ties -> index order must decide
    if rows > 1:
        x[1, :] = 0.5                                 # all equal: indices 0..k-1
    vals = np.empty((rows, k), np.float32)
    idx = np.empty((rows, k), np.int32)
    L.check(eng._h, L.lib.asr_op_log_softmax_topk(eng._h, L.fptr(x), rows, vocab, k, L.fptr(vals), L.iptr(idx)))
    z32 = O.log_softmax(x)
    wv, wi = O.top_k(z32, k)
    z64 = O.log_softmax(x.astype(np.float64))
    # candidate values within float32 rounding of the float64 log-softmax
    assert np.max(np.abs(vals - np.take_along_axis(z64, idx.astype(np.int64), axis=1))) < 5e-6
    # indices: identical to the oracle wherever the float32 oracle has no near-tie at the cut
    assert np.array_equal(idx[:2], wi[:2])            # the crafted exact-tie rows
    srt = np.sort(z64, axis=1)[:, ::-1]
    gap_ok = np.ones(rows, bool)
    for r in range(rows):
        top = srt[r, :min(k + 1, vocab)]
        gap_ok[r] = np.all(np.abs(np.diff(top)) > 1e-5) if len(top) > 1 else True
    assert np.array_equal(idx[gap_ok], wi[gap_ok])
    assert gap_ok.mean() > 0.5 or rows < 8


@pytest.mark.parametrize('b,k', [(64, 10), (3, 3), (5, 1), (2, 13), (1, 32)])
def test_beam_step_bit_exact(L, eng, b, k):
    rng = np.random.default_rng(b * 100 + k)
    R = b * k
    scores = (-rng.random(R) * 20).astype(np.float32)
    scores[::k] = 0.0
    fin = (rng.random(R) < 0.3)
    if R > 2:
        scores[1] = -1e10                              # a dead beam (model.py:264)
    cs = np.sort((-rng.random((R, k)) * 8).astype(np.float32), axis=1)[:, ::-1].copy()
    ct = rng.integers(0, 50, size=(R, k)).astype(np.int32)
    ct[rng.random((R, k)) < 0.15] = 3
    if k > 1:
        cs[0, 1] = cs[0, 0]                            # exact tie inside a row -> lower flat index first
    ns = np.empty(R, np.float32)
    tok = np.empty(R, np.int32)
    par = np.empty(R, np.int32)
    fo = np.empty(R, np.int32)
    L.check(eng._h, L.lib.asr_op_beam_step(eng._h, L.fptr(scores), L.iptr(fin.astype(np.int32)), L.fptr(cs), L.iptr(ct), b, k,
                                           L.fptr(ns), L.iptr(tok), L.iptr(par), L.iptr(fo)))
    ws, wt, wp, wf = O.beam_bookkeeping(scores, fin, cs, ct, k)
    assert np.array_equal(ns, ws)
    assert np.array_equal(tok, wt)
    assert np.array_equal(par, wp)
    assert np.array_equal(fo.astype(bool), wf)


@pytest.mark.parametrize('rows,vocab,position,scale', [(7, 50, 0, 1), (40, 300, 13, 1), (3, 11, 200, 0), (1, 2, 511, 1)])
def test_embedding_plus_timing_signal_bit_exact(L, eng, rows, vocab, position, scale):
    """a14 (model.py:569-576): y = E[token] * sqrt(H) + signal[position]; one multiply and one add in float32, so the
    decode step's kernel must give the numpy float32 evaluation of the oracle's expressions bit for bit."""
    hidden = eng.hidden
    rng = np.random.default_rng(rows * 31 + position)
    E = rng.standard_normal((vocab, hidden)).astype(np.float32)
    tok = rng.integers(0, vocab, rows).astype(np.int32)
    tok[0] = vocab - 1
    y = np.empty((rows, hidden), np.float32)
    L.check(eng._h, L.lib.asr_op_embedding(eng._h, L.fptr(E), L.iptr(tok), rows, vocab, hidden, position, scale, L.fptr(y)))
    cfg = tiny_config(hidden_units=hidden, scale_embedding=bool(scale))
    want = O.embed(tok, {'decoder/dst_embedding/kernel': E}, cfg, np.float32) + O.timing_signal(position + 1, hidden)[position]
    assert want.dtype == np.float32
    assert np.array_equal(y, want)
    # a token id outside the table is refused, not read
    bad = tok.copy()
    bad[-1] = vocab
    assert L.lib.asr_op_embedding(eng._h, L.fptr(E), L.iptr(bad), rows, vocab, hidden, position, scale, L.fptr(y)) != 0


@pytest.mark.parametrize('n,t,heads', [(1, 1, 2), (3, 77, 2), (2, 334, 16), (5, 64, 4)])
def test_pack_cross_kv_is_the_head_major_copy(L, eng, n, t, heads):
    """kv [n*t, 2H] (k | v) -> k, v [n][heads][t][64]: a pure copy in float32 (bit-exact against the numpy transpose,
    = split_heads of utils.py:618-619), round-to-nearest-even bf16 for the bf16 caches of `precision: bf16`."""
    H = heads * 64
    rng = np.random.default_rng(n * 1000 + t)
    kv = rng.standard_normal((n * t, 2 * H)).astype(np.float32)
    kv[0, :4] = [1.00390625, -1.00390625, 1.01171875, 3.0e-41]        # bf16 ties (to even, both signs), a subnormal
    want_k = kv[:, :H].reshape(n, t, heads, 64).transpose(0, 2, 1, 3)
    want_v = kv[:, H:].reshape(n, t, heads, 64).transpose(0, 2, 1, 3)
    k = np.empty((n, heads, t, 64), np.float32)
    v = np.empty_like(k)
    L.check(eng._h, L.lib.asr_op_pack_cross_kv(eng._h, L.fptr(kv), n, t, heads, 0, k.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p)))
    assert np.array_equal(k, want_k) and np.array_equal(v, want_v)
    k16 = np.empty((n, heads, t, 64), np.uint16)
    v16 = np.empty_like(k16)
    L.check(eng._h, L.lib.asr_op_pack_cross_kv(eng._h, L.fptr(kv), n, t, heads, 1, k16.ctypes.data_as(C.c_void_p), v16.ctypes.data_as(C.c_void_p)))

    def bf16_rne(x):
        u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
        return ((u + 0x7fff + ((u >> 16) & 1)) >> 16).astype(np.uint16)
    assert np.array_equal(k16, bf16_rne(want_k)) and np.array_equal(v16, bf16_rne(want_v))


def _bf16_round(x):
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    return (((u + 0x7fff + ((u >> 16) & 1)) >> 16) << 16).astype(np.uint32).view(np.float32).reshape(x.shape)


@pytest.mark.parametrize('variant,tol', [(0, 2e-6), (1, 2e-6), (2, 3e-5)])
@pytest.mark.parametrize('n,t,heads,beams', [(1, 1, 2, 1), (3, 37, 2, 10), (2, 334, 16, 10), (2, 100, 4, 20), (4, 16, 3, 16), (2, 1000, 2, 13)])
def test_decoder_cross_attention_kernel_variants(L, eng, n, t, heads, beams, variant, tol):
    """The decode-time encoder-decoder attention kernel alone (a8/a9: model.py:604-616 -> utils.py:612-631; -1e9 padding
    bias of common_attention.py:202), all three variants the engine launches, against the float64 evaluation of the
    oracle's expressions on the same operands: exact-f32 products and the default mode's three fp16 products per multiply
    to the dense layers' bar (2e-6, max-abs / max-abs); the `precision: bf16` variant against the SAME float64 expressions
    on K / V rounded to bf16 as its cache holds them - its own arithmetic adds only q and P at 16 significant bits (3e-5),
    far below the mode's stated 5e-2.  Shapes: one key, ragged key counts around the 16-key blocks, two beam groups
    (beams > 16), T = 1000 (config 5), padded positions incl. a whole padded tail.  Observed (max-abs / max-abs): exact f32
    3e-7 .. 1e-6, fp16x3 3e-7 .. 5e-7, bf16 K/V 4e-6 .. 7e-6."""
    H = heads * 64
    rng = np.random.default_rng(1000 * n + t + beams + variant)
    q = rng.standard_normal((n * beams, H)).astype(np.float32) * 2.0
    qb = (rng.standard_normal(H) * 0.1).astype(np.float32)
    kv = rng.standard_normal((n * t, 2 * H)).astype(np.float32)
    pad = np.zeros((n, t), np.int32)
    for u in range(n):
        pad[u, t - (u * t) // (2 * n):] = 1                         # utterance u: a padded tail (none for u = 0)
    if t > 5:
        pad[n - 1, 2] = 1                                           # and one padded position in the middle
    ctx = np.empty((n * beams, H), np.float32)
    flag = np.zeros(1, np.int32)
    L.check(eng._h, L.lib.asr_op_decoder_cross_attention(eng._h, L.fptr(q), L.fptr(qb), L.fptr(kv), L.iptr(pad.reshape(-1)), n, t, heads,
                                                         beams, variant, L.fptr(ctx), L.iptr(flag)))
    assert flag[0] == 0
    kv_used = _bf16_round(kv) if variant == 2 else kv
    k64 = O.split_heads(kv_used[:, :H].astype(np.float64).reshape(n, t, H), heads)
    v64 = O.split_heads(kv_used[:, H:].astype(np.float64).reshape(n, t, H), heads)
    q64 = O.split_heads((q.astype(np.float64) + qb.astype(np.float64)).reshape(n, beams, H), heads) * 64 ** -0.5
    bias = (pad.astype(np.float64) * -1e9)[:, None, None, :]
    want = O.combine_heads(O.dot_product_attention(q64, k64, v64, bias)).reshape(n * beams, H)
    assert rel_err(ctx, want) < tol
    # per-row reading of the same bar (rows with small outputs are not hidden behind the largest one)
    row_err = np.abs(ctx - want).max(axis=1) / np.abs(want).max(axis=1)
    assert row_err.max() < 10 * tol


def test_decoder_cross_attention_f16x3_variant_reports_operands_outside_fp16_range(L, eng):
    """Variant 1 multiplies fp16 pieces: a key (checked where K / V are laid out) or a query (checked where q is assembled)
    beyond fp16's range must raise the flag the engine repeats the call on - never a silently wrong context."""
    n, t, heads, beams = 2, 40, 2, 10
    H = heads * 64
    rng = np.random.default_rng(5)
    q = rng.standard_normal((n * beams, H)).astype(np.float32)
    qb = np.zeros(H, np.float32)
    kv = rng.standard_normal((n * t, 2 * H)).astype(np.float32)
    pad = np.zeros(n * t, np.int32)
    ctx = np.empty((n * beams, H), np.float32)
    for which in ('k', 'q'):
        q2, kv2 = q.copy(), kv.copy()
        if which == 'k':
            kv2[17, 5] = 1.0e5
        else:
            q2[3, 70] = 9.0e5                                       # x 1/8 after the bias: still beyond 65504
        flag = np.zeros(1, np.int32)
        L.check(eng._h, L.lib.asr_op_decoder_cross_attention(eng._h, L.fptr(q2), L.fptr(qb), L.fptr(kv2), L.iptr(pad), n, t, heads, beams, 1,
                                                             L.fptr(ctx), L.iptr(flag)))
        assert flag[0] == 1, which


@pytest.mark.parametrize('kv_bf16,tol', [(0, 2e-6), (1, 3e-5), (2, 2e-6)])
@pytest.mark.parametrize('n,pos,heads,beams', [(2, 0, 2, 10), (1, 1, 2, 1), (3, 7, 2, 10), (2, 40, 4, 10), (2, 33, 2, 20), (2, 199, 2, 10),
                                               (1, 16, 3, 16)])
def test_decoder_self_attention_kernel_variants(L, eng, n, pos, heads, beams, kv_bf16, tol):
    """The decode-time self-attention kernel alone (a8/a9/a16: model.py:588-600, utils.py:612-631 with reserve_last) against
    the float64 evaluation of the oracle's expressions, beam by beam over that beam's OWN history: the reference gathers
    its cache with the surviving beams' indices at every step (model.py:296-301); the kernel leaves rows where they were
    written and follows an ancestry table, streaming each LIVE (position, slot) row once for all beams - so the ancestry
    here comes from simulated random survivor choices (beams share prefixes, some rows are dead).  float32 caches to the
    dense layers' bar, the `precision: bf16` variant against the same expressions on rows rounded to bf16 as its cache
    holds them (q and P at 16 significant bits on top: 3e-5).  Position 0 (no history), one beam, two beam groups
    (beams > 16), 199 positions (the last step of max_target_length 200)."""
    H, K = heads * 64, beams
    rng = np.random.default_rng(100 * n + pos + beams + kv_bf16)
    qkv = rng.standard_normal((n * K, 3 * H)).astype(np.float32)
    qkv[:, :H] *= 2.0
    bias = (rng.standard_normal(3 * H) * 0.1).astype(np.float32)
    hk = rng.standard_normal((n, pos, K, H)).astype(np.float32)
    hv = rng.standard_normal((n, pos, K, H)).astype(np.float32)
    anc = np.zeros((n, K, pos), np.int32)
    for u in range(n):
        a = np.zeros((K, 0), np.int32)
        for j in range(pos):
            a = np.concatenate([a, np.arange(K, dtype=np.int32)[:, None]], axis=1)     # step j: beam b writes slot b ...
            a = a[np.sort(rng.integers(0, max(1, K // 2 + 1), K))] if j % 3 else a[rng.integers(0, K, K)]   # ... survivors are chosen
        anc[u] = a
    ctx = np.empty((n * K, H), np.float32)
    L.check(eng._h, L.lib.asr_op_decoder_self_attention(eng._h, L.fptr(qkv), L.fptr(bias), L.fptr(hk.reshape(-1)), L.fptr(hv.reshape(-1)),
                                                        L.iptr(anc.reshape(-1)), n, pos, heads, K, kv_bf16, L.fptr(ctx)))
    x = qkv + bias                                                  # float32, as the kernel adds it
    rnd = _bf16_round if kv_bf16 == 1 else (lambda z: z)           # (variant 2: float32 caches, fp16x3 products)
    q = x[:, :H].astype(np.float64).reshape(n, K, H) * 64 ** -0.5
    kn = rnd(np.ascontiguousarray(x[:, H:2 * H])).astype(np.float64).reshape(n, K, H)
    vn = rnd(np.ascontiguousarray(x[:, 2 * H:])).astype(np.float64).reshape(n, K, H)
    hk64, hv64 = rnd(hk).astype(np.float64), rnd(hv).astype(np.float64)
    want = np.empty((n, K, H))
    jj = np.arange(pos)
    for u in range(n):
        for b in range(K):
            keys = np.concatenate([hk64[u, jj, anc[u, b]], kn[u, b][None]], axis=0)          # [pos + 1, H]
            vals = np.concatenate([hv64[u, jj, anc[u, b]], vn[u, b][None]], axis=0)
            k4 = O.split_heads(keys[None], heads)
            v4 = O.split_heads(vals[None], heads)
            q4 = O.split_heads(q[u, b][None, None], heads)
            want[u, b] = O.combine_heads(O.dot_product_attention(q4, k4, v4, None))[0, 0]
    want = want.reshape(n * K, H)
    assert rel_err(ctx, want) < tol
    row_err = np.abs(ctx - want).max(axis=1) / np.abs(want).max(axis=1)
    assert row_err.max() < 10 * tol
