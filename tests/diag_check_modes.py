"""(Test infrastructure, not collected by pytest: it imports the CPU oracle, which only tests/, bench.py's CPU leg and smoke() may.)
One-off diagnostic (GPU box): decode bench.py's 64-utterance batch in every precision mode and with the CPU oracle
(float32, K/V-cache algorithm, and float64 on the utterances where anything differs), and say per mode how many
utterances are token-identical to the oracle and where the first difference lies.  Random-init weights over 201 steps
leave near-ties; this tells a near-tie from a defect.

    python tests/diag_check_modes.py [n_utts] > gpurun_out/check_modes.json
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from oracle import asr_oracle as O  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    from asr_amd.model import Engine
    cfg = bench.bench_config(10, 200, 'f32_f16x3', 3900)
    ocfg = bench._oracle_config(cfg)
    from asr_amd import weights
    W = weights.random_init(cfg, seed=0)                        # bench.py's weights and rank-0 batch
    raw = bench.synthetic_raw(n, 1000, seed=1000)
    X = O.make_batch([raw[i] for i in range(n)])
    out = {'n': n}
    t0 = time.time()
    Y32 = O.build_test_model(X, W, ocfg, num_gpus=1, algo='kv')
    out['oracle_f32_s'] = round(time.time() - t0, 1)
    print('oracle float32 done in %.0f s' % (time.time() - t0), file=sys.stderr, flush=True)
    res = {}
    for prec in ('f32_f16x3', 'f32_bf16x6', 'f32', 'bf16'):
        e = Engine(bench.bench_config(10, 200, prec, 3900), 0)
        e.load_weights(W)
        Y = e.decode(X)
        e.close()
        m = min(Y.shape[1], Y32.shape[1])
        same = np.all(Y[:, :m] == Y32[:, :m], axis=1)
        diff = [int(u) for u in np.nonzero(~same)[0]]
        res[prec] = {'identical_to_oracle_f32': int(same.sum()), 'differ': diff,
                     'first_diff_step': {str(u): int(np.nonzero(Y[u, :m] != Y32[u, :m])[0][0]) for u in diff}}
        res[prec]['_Y'] = Y
        print(prec, res[prec]['identical_to_oracle_f32'], diff, file=sys.stderr, flush=True)
    # float64 oracle on the utterances where any f32-accurate mode differs from the float32 oracle
    suspects = sorted(set(sum([res[p]['differ'] for p in ('f32_f16x3', 'f32_bf16x6', 'f32')], [])))[:6]
    out['float64_check'] = {}
    W64 = O.cast_weights(W, np.float64)
    for u in suspects:
        Xu = O.make_batch([raw[u]]).astype(np.float64)
        Y64 = O.build_test_model(Xu, W64, ocfg, num_gpus=1, algo='kv')
        row = {}
        for name, Yx in [('oracle_f32', Y32)] + [(p, res[p]['_Y']) for p in res]:
            m = min(Yx.shape[1], Y64.shape[1])
            row[name] = bool(np.array_equal(Yx[u, :m], Y64[0, :m]))
        out['float64_check'][str(u)] = row
        print('utt', u, row, file=sys.stderr, flush=True)
    for p in res:
        del res[p]['_Y']
    out['modes'] = res
    print(json.dumps(out))


if __name__ == '__main__':
    main()
