"""The two independent CPU restatements of the reference path must agree.

oracle/asr_oracle.py (numpy; 'ref' and 'kv' decode algorithms) and oracle/asr_oracle_torch.py (torch-CPU; the
reference's own data flow, written from the reference source without the numpy file) are both PARITY UNPINNED
against TensorFlow (no TF, no fixtures in the reference); agreement between them only rules out a misreading that
one of them made alone.  Tolerances: encoder output 1e-5 (max-abs / max-abs), logits 1e-4 in float32, tokens
identical."""
import numpy as np
import pytest
import torch

from oracle import asr_oracle as O
from oracle import asr_oracle_torch as T
from tests.helpers import rel_err, tiny_config, tiny_inputs

CASES = [
    dict(name='beam3_untied_glu', cfg=dict(), lens=(60, 90, 75), seed=1, bias_scale=0.1),
    dict(name='greedy', cfg=dict(beam_size=1), lens=(60, 90, 75, 30), seed=2, bias_scale=0.1),
    dict(name='beam10_tied', cfg=dict(beam_size=10, tie_embedding_and_softmax=True, max_target_length=12), lens=(100, 40), seed=3,
         bias_scale=0.2),
    dict(name='relu_short_T', cfg=dict(beam_size=2, ff_activation='relu', max_target_length=200), lens=(9, 6), seed=5, bias_scale=0.1),
    dict(name='no_embedding_scale', cfg=dict(beam_size=4, scale_embedding=False, max_target_length=10), lens=(33, 50), seed=6,
         bias_scale=0.1),
]


def test_timing_signal_bit_identical():
    for length, ch in ((1, 128), (40, 128), (334, 1024)):
        a = O.timing_signal(length, ch)
        b = T.timing_signal(length, ch, torch.float32).numpy()[0]
        assert a.dtype == np.float32 and np.array_equal(a, b)
    s = T.timing_signal(3, 1024, torch.float32).numpy()[0]
    assert np.all(s[0, :512] == 0) and np.all(s[0, 512:] == 1)            # SURVEY 8c known answers
    assert abs(float(s[1, 0]) - 0.8414709848) < 1e-7


@pytest.mark.parametrize('case', CASES, ids=[c['name'] for c in CASES])
def test_numpy_and_torch_restatements_agree(case):
    ocfg = tiny_config(**case['cfg'])
    W = O.random_weights(ocfg, seed=case['seed'], bias_scale=case['bias_scale'])
    X = tiny_inputs(case['lens'], seed=case['seed'] + 10)
    enc_np, pad_np = O.encoder(X, W, ocfg)
    enc_t, pad_t = T.encoder(X, W, ocfg, torch.float32)
    assert np.array_equal(pad_np, pad_t.numpy())
    assert rel_err(enc_t.numpy(), enc_np) < 1e-5
    enc64, _ = T.encoder(X, W, ocfg, torch.float64)
    assert rel_err(enc_np, enc64.numpy()) < 1e-5
    # decode: literal reference data flow (torch) vs both numpy algorithms
    tr_t, tr_kv = {}, {}
    Yt = T.beam_search(enc_np, W, ocfg, torch.float32, trace=tr_t)
    Ykv = O.beam_search(enc_np, W, ocfg, algo='kv', trace=tr_kv)
    Yref = O.beam_search(enc_np, W, ocfg, algo='ref')
    assert Yt.shape == Ykv.shape and np.array_equal(Yt, Ykv) and np.array_equal(Yt, Yref)
    assert len(tr_t['tokens']) == len(tr_kv['tokens'])
    for s in range(len(tr_t['tokens'])):
        assert np.array_equal(tr_t['tokens'][s], tr_kv['tokens'][s]), s
        assert np.array_equal(tr_t['parents'][s], tr_kv['parents'][s]), s
        assert rel_err(tr_t['logits'][s], tr_kv['logits'][s]) < 1e-4
    # float64 evaluation of the torch restatement picks the same hypotheses (no near-tie decides the case)
    Y64 = T.beam_search(enc64.numpy(), O.cast_weights(W, np.float64), ocfg, torch.float64)
    assert np.array_equal(Y64, Yt)


def test_tower_sharding_agrees():
    ocfg = tiny_config(beam_size=2, max_target_length=15)
    W = O.random_weights(ocfg, seed=7, bias_scale=0.1)
    X = tiny_inputs((90, 60, 30, 45, 20), seed=8)
    for towers in (1, 2, 7):
        assert np.array_equal(T.build_test_model(X, W, ocfg, towers), O.build_test_model(X, W, ocfg, num_gpus=towers, algo='kv'))


def test_golden_fixture_matches_the_torch_restatement(repo_root):
    """tests/golden/tiny_decode.json was frozen from the numpy oracle; the torch restatement reproduces it."""
    import json
    import os
    g = json.load(open(os.path.join(repo_root, 'tests', 'golden', 'tiny_decode.json')))
    for c in g['cases']:
        ocfg = tiny_config(**c['cfg'])
        W = O.random_weights(ocfg, seed=c['seed'], bias_scale=c['bias_scale'], emb_scale=c['emb_scale'])
        X = tiny_inputs(tuple(c['lens']), seed=c['seed'] + 10)
        assert T.build_test_model(X, W, ocfg, c['towers']).tolist() == c['Y'], c['name']
