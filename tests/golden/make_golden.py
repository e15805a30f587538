"""Generates tests/golden/tiny_decode.json from the CPU oracle (oracle/asr_oracle.py).

The reference (Python 2 + TensorFlow 1.x) cannot be imported or run in the build container and ships
no fixtures, so these vectors are ORACLE output, not reference output: they freeze the oracle and give
the GPU tests a fixture that needs no oracle run.  Usage:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import asr_oracle as O          # noqa: E402
from tests.helpers import tiny_config, tiny_inputs   # noqa: E402

CASES = [
    dict(name='beam3_untied', cfg=dict(), lens=[60, 90, 75], seed=1, bias_scale=0.1, emb_scale=1.0, towers=1),
    dict(name='greedy_untied', cfg=dict(beam_size=1), lens=[60, 90, 75, 30], seed=2, bias_scale=0.1, emb_scale=1.0, towers=1),
    dict(name='beam10_tied', cfg=dict(beam_size=10, tie_embedding_and_softmax=True, max_target_length=25), lens=[100, 40],
         seed=3, bias_scale=0.2, emb_scale=1.0, towers=1),
    dict(name='beam2_three_towers', cfg=dict(beam_size=2, max_target_length=15), lens=[90, 60, 30, 45, 20], seed=7,
         bias_scale=0.1, emb_scale=1.0, towers=3),
]

out = {'generator': 'tests/golden/make_golden.py', 'source': 'oracle/asr_oracle.py (float32, K/V-cache decode)', 'cases': []}
for c in CASES:
    ocfg = tiny_config(**c['cfg'])
    W = O.random_weights(ocfg, seed=c['seed'], bias_scale=c['bias_scale'], emb_scale=c['emb_scale'])
    X = tiny_inputs(tuple(c['lens']), seed=c['seed'] + 10)
    Y = O.build_test_model(X, W, ocfg, c['towers'], 'kv')
    Y64 = O.build_test_model(X, W, ocfg, c['towers'], 'ref', np.float64)
    assert np.array_equal(Y, Y64), c['name']
    enc, _ = O.encoder(X, W, ocfg)
    d = dict(c)
    d['Y'] = Y.tolist()
    d['enc_abs_sum'] = float(np.abs(enc).sum())
    out['cases'].append(d)
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tiny_decode.json'), 'w') as f:
    json.dump(out, f, indent=1)
print('wrote', len(out['cases']), 'cases')
