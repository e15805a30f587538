"""CPU checks of the operand-splitting arithmetic behind the default GEMM modes (asr_amd/csrc/gemm_split.hip):

* f32_bf16x6: x = h + m + l with three bf16 terms is EXACT, and the six products kept reproduce a float64 dot product
  to f32-chain accuracy;
* f32_f16x3: x = x1 + 2^-11 x2 with two fp16 terms carries 22 significand bits inside fp16's normal range, and three
  products in two f32 accumulators are at least as accurate as an f32 BLAS dot product.

These emulate the kernels' arithmetic in numpy (same splits, same accumulator structure, k-tiles of 32); the GPU kernels
themselves are held to the same bars in tests/test_gpu_precision.py.
"""
import numpy as np


def bf16_rne(x):
    u = np.ascontiguousarray(x, np.float32).view(np.uint32)
    r = (u + np.uint32(0x7fff) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xffff0000)
    return r.view(np.float32)


def split_bf16(x):
    h = bf16_rne(x)
    m = bf16_rne(x - h)
    l = bf16_rne(x - h - m)
    return h, m, l


def split_f16(x):
    x1 = x.astype(np.float16)
    x2 = ((x - x1.astype(np.float32)) * np.float32(2048.0)).astype(np.float16)
    return x1, x2


def chain(acc, prod64):
    """acc (f32) += product block computed exactly (float64), rounded to f32 once per k-tile like an MFMA accumulate."""
    return (acc.astype(np.float64) + prod64).astype(np.float32)


def gemm_bf16x6(A, W, kt=32):
    Ah, Am, Al = [t.astype(np.float64) for t in split_bf16(A)]
    Wh, Wm, Wl = [t.astype(np.float64) for t in split_bf16(W)]
    acc = np.zeros((A.shape[0], W.shape[0]), np.float32)
    for k0 in range(0, A.shape[1], kt):
        s = slice(k0, k0 + kt)
        for a, w in ((Al, Wh), (Ah, Wl), (Am, Wm), (Am, Wh), (Ah, Wm), (Ah, Wh)):      # smallest first, as the kernel does
            acc = chain(acc, a[:, s] @ w[:, s].T)
    return acc


def gemm_f16x3(A, W, kt=32):
    a1, a2 = [t.astype(np.float64) for t in split_f16(A)]
    w1, w2 = [t.astype(np.float64) for t in split_f16(W)]
    hi = np.zeros((A.shape[0], W.shape[0]), np.float32)
    lo = np.zeros_like(hi)
    for k0 in range(0, A.shape[1], kt):
        s = slice(k0, k0 + kt)
        lo = chain(lo, a1[:, s] @ w2[:, s].T)
        lo = chain(lo, a2[:, s] @ w1[:, s].T)
        hi = chain(hi, a1[:, s] @ w1[:, s].T)
    return hi + lo * np.float32(1.0 / 2048.0)


def test_bf16_three_term_split_is_exact():
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(100000) * np.exp(rng.uniform(-20, 20, 100000))).astype(np.float32)
    h, m, l = split_bf16(x)
    assert np.array_equal((h.astype(np.float64) + m.astype(np.float64) + l.astype(np.float64)).astype(np.float32), x)
    assert np.array_equal(h.astype(np.float64) + m + l, x.astype(np.float64))          # exact even before rounding back


def test_f16_two_term_split_carries_22_bits_in_range():
    rng = np.random.default_rng(1)
    mag = np.exp(rng.uniform(np.log(2.0 ** -10), np.log(60000.0), 100000))
    x = (mag * rng.choice([-1.0, 1.0], mag.size)).astype(np.float32)
    x1, x2 = split_f16(x)
    back = x1.astype(np.float64) + x2.astype(np.float64) / 2048.0
    assert np.all(np.isfinite(x1.astype(np.float32)))
    assert np.max(np.abs(back - x) / np.abs(x)) <= 2.0 ** -22
    # outside the range the split is NOT usable - which is why the kernels flag such operands and the call is redone on bf16x6
    assert not np.isfinite(np.float32(70000.0).astype(np.float16))
    tiny = np.float32(3.0e-7)
    t1, t2 = split_f16(np.array([tiny]))
    assert abs((float(t1[0]) + float(t2[0]) / 2048.0) - float(tiny)) / float(tiny) > 2.0 ** -22


def test_split_gemms_are_as_accurate_as_f32_blas():
    rng = np.random.default_rng(2)
    for (m, n, k, wscale) in ((96, 160, 1024, 0.05), (64, 128, 4096, 0.03)):
        A = rng.standard_normal((m, k)).astype(np.float32)
        W = rng.uniform(-wscale, wscale, (n, k)).astype(np.float32)
        ref = A.astype(np.float64) @ W.astype(np.float64).T
        rms = lambda y: float(np.sqrt(np.mean((y.astype(np.float64) - ref) ** 2)))
        e_blas = rms(A @ W.T)
        e_x6 = rms(gemm_bf16x6(A, W))
        e_f16 = rms(gemm_f16x3(A, W))
        scale = float(np.sqrt(np.mean(ref ** 2)))
        assert e_x6 <= 1.5 * e_blas and e_f16 <= 1.5 * e_blas, (e_blas, e_x6, e_f16)
        assert e_x6 / scale < 1e-6 and e_f16 / scale < 1e-6
