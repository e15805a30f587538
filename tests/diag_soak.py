"""Diagnostic (not collected by pytest; run by hand on the GPU box): many random tiny models / batches decoded by the library in
the default mode and compared with the float32 and float64 oracles.  Prints the cases where the GPU differs from the float32
oracle, and whether the two oracles agree there (if they do not, the case has a float32 near-tie of its own).
    python tests/diag_soak.py [cases]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import asr_oracle as O  # noqa: E402
from tests.helpers import product_config, tiny_config, tiny_inputs  # noqa: E402


def main():
    from asr_amd.model import Engine
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(2026)
    bad = ties = 0
    for c in range(cases):
        beam = int(rng.choice([1, 2, 3, 5, 10, 13, 20]))
        mtl = int(rng.choice([6, 12, 25, 40]))
        n = int(rng.integers(1, 9))
        lens = tuple(int(x) for x in rng.integers(20, 140, n))
        ocfg = tiny_config(beam_size=beam, max_target_length=mtl)
        W = O.random_weights(ocfg, seed=1000 + c, bias_scale=0.1)
        X = tiny_inputs(lens, seed=2000 + c)
        shared = int(rng.integers(0, 2))
        cfg = product_config(ocfg)
        cfg.test['shared_gpu'] = shared
        eng = Engine(cfg, 0)
        eng.load_weights(W)
        try:
            Y = eng.decode(X)
        finally:
            eng.close()
        want = O.build_test_model(X, W, ocfg, num_gpus=1, algo='kv')
        if Y.shape == want.shape and np.array_equal(Y, want):
            continue
        w64 = O.build_test_model(X.astype(np.float64), O.cast_weights(W, np.float64), ocfg, num_gpus=1, algo='kv')
        agree = w64.shape == want.shape and np.array_equal(w64, want)
        gpu64 = w64.shape == Y.shape and np.array_equal(w64, Y)
        bad += 1
        ties += 0 if agree else 1
        print('case %d (beam %d, mtl %d, n %d, shared %d): GPU != float32 oracle; oracles agree: %s; GPU == float64 oracle: %s'
              % (c, beam, mtl, n, shared, agree, gpu64), flush=True)
    print('%d cases, %d differ from the float32 oracle, of which %d where the two oracles disagree themselves' % (cases, bad, ties))


if __name__ == '__main__':
    main()
