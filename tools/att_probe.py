import sys, numpy as np
sys.path.insert(0, '/root/repo')
from oracle import asr_oracle as O
from tests.helpers import product_config
from asr_amd import _lib as L
from asr_amd.model import Engine
import time
for prec in ('f32_f16x3', 'f32', 'bf16'):
    ocfg = O.make_config(beam_size=2, max_target_length=4)
    eng = Engine(product_config(ocfg, precision=prec), 0)
    W = O.random_weights(ocfg, seed=1); eng.load_weights(W)
    hidden, heads = eng.hidden, eng.cfg.num_heads
    for (n, t, lens) in [(2, 70, (70, 41)), (3, 334, (334, 200, 17)), (1, 1000, (900,))]:
        rng = np.random.default_rng(t)
        qkv = rng.standard_normal((n, t, 3 * hidden)).astype(np.float32)
        pad = np.zeros((n, t), np.int32)
        for i, l in enumerate(lens): pad[i, l:] = 1
        ctx = np.empty((n, t, hidden), np.float32)
        L.check(eng._h, L.lib.asr_op_encoder_attention(eng._h, L.fptr(qkv), L.iptr(pad), n, t, L.fptr(ctx)))
        q, k, v = np.split(qkv.astype(np.float64), 3, axis=2)
        q = O.split_heads(q, heads) * 64 ** -0.5
        bias = (pad.astype(np.float64) * -1e9)[:, None, None, :]
        want = O.combine_heads(O.dot_product_attention(q, O.split_heads(k, heads), O.split_heads(v, heads), bias))
        err = np.max(np.abs(ctx - want)) / np.max(np.abs(want))
        print(prec, n, t, 'max err / max', err, 'retries', eng.range_retries(), flush=True)
    eng.close()
