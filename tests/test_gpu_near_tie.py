"""north_star asks for the reference's beam sequences on the fixed synthetic set.  On bench.py's own batch (64 x 10 s,
random-init char model, 201 steps) a float32 implementation cannot promise that literally: random-init posteriors are
nearly flat, beam scores reach -600 after 100 steps, and now and then two candidate totals of the pruning step
(model.py:284-301) lie closer together than ONE float32 ulp of the score - which of them survives then depends on the
order a float32 sum was taken in (tile shapes, K-slices), in TensorFlow as much as here.  This test is the collected form
of round 2's diagnostics (tests/diag_near_tie.py, profiles/r02_check_modes.json): the GPU decodes the WHOLE 64-utterance
batch with the kernel shapes of the headline (shared-GPU regime, one run of 640-row kernels) in the default mode and in
exact f32; for utterances 0-5 and 14 (4 and 14 are the two that differed in round 2) every hypothesis must either equal
the float32 oracle's, or the test PROVES that the departure is a tie and not a defect:
  * the first step at which the GPU's (token, parent) rows leave the float64 oracle's is found from the per-step traces;
  * at that step the float64 oracle's own candidate totals around the pruning boundary are less than 2 float32 ulps of the
    score apart (a defect would show as a departure at a step with a clear gap), and up to that step the GPU's beam scores
    follow the float64 ones to float32 summation accuracy;
  * decoded alone (other tile shapes, other summation order) the utterance again either equals the oracle or leaves it at
    a proven tie."""
import os
import sys

import numpy as np
import pytest

from oracle import asr_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

UTTS = [0, 1, 2, 3, 4, 5, 14]
K = 10


def _f64_trace(ocfg, W64, raw_u):
    X = O.make_batch([raw_u]).astype(np.float64)
    enc, _ = O.encoder(X, W64, ocfg)
    tr = {}
    Y = O.beam_search(enc, W64, ocfg, algo='kv', trace=tr)
    return Y, tr


def _departure(tokens, parents, scores, tr64, row0):
    """First step at which the GPU's K rows of one utterance (rows row0 .. row0+K of the batch trace) differ from the float64
    oracle's single-utterance trace; parents are batch-global row indices on the GPU side.  -> (step or None, evidence)."""
    steps = min(len(tr64['tokens']), tokens.shape[0])
    for s in range(steps):
        gt = tokens[s, row0:row0 + K]
        gp = parents[s, row0:row0 + K] - row0
        if np.array_equal(gt, tr64['tokens'][s]) and np.array_equal(gp, tr64['parents'][s]):
            continue
        prev = tr64['scores'][s - 1] if s > 0 else np.array([0.0] + [-1e10] * (K - 1))
        cs, _ = O.top_k(O.log_softmax(tr64['logits'][s]), K)
        tot = np.sort((prev[:, None] + cs).reshape(-1))[::-1][:K + 1]         # the K survivors and the first one pruned
        gaps = tot[:-1] - tot[1:]
        ulp = float(np.spacing(np.float32(abs(tot[K - 1]))))
        errs = [float(np.max(np.abs(scores[j, row0:row0 + K] - tr64['scores'][j])[tr64['scores'][j] > -1e9])) for j in range(s)]
        return s, {'step': s, 'min_gap': float(gaps.min()), 'ulp_of_score': ulp, 'score': float(tot[K - 1]),
                   'max_score_err_before': max(errs) if errs else 0.0}
    return None, None


def _assert_tie(ev, what):
    # the float64 oracle's own candidates are closer than 2 float32 ulps of the score: either order is a correct float32 answer
    assert ev['min_gap'] < 2.0 * ev['ulp_of_score'], (what, ev)
    # ... and up to there the GPU followed the float64 scores as closely as a float32 sum of that many terms can
    assert ev['max_score_err_before'] < 64.0 * ev['ulp_of_score'], (what, ev)


@pytest.mark.parametrize('precision', ['f32_f16x3', 'f32'])
def test_headline_batch_hypotheses_equal_the_oracle_or_leave_it_at_a_proven_tie(precision):
    import bench
    from asr_amd import weights
    from asr_amd.model import Engine
    cfg = bench.bench_config(K, 200, precision, 3900)
    ocfg = bench._oracle_config(cfg)
    W = weights.random_init(cfg, seed=0)                           # bench.py's weights and its rank-0 batch
    raw = bench.synthetic_raw(64, 1000, seed=1000)
    X64 = O.make_batch([raw[i] for i in range(64)])
    # float32 oracle on the sample: all utterances have the same length, so a batch of seven pads (Q1) exactly as the batch of 64
    Y32 = O.build_test_model(O.make_batch([raw[u] for u in UTTS]), W, ocfg, num_gpus=1, algo='kv')
    eng = Engine(cfg, 0)
    eng.load_weights(W)
    try:
        eng.set_sharing(1, 1)                                       # the headline's regime: 640-row kernels, fewer K-slices
        Y, tg = eng.decode_trace(X64, 201, logits=False)
        assert Y.shape == (64, 201)
        m = min(Y.shape[1], Y32.shape[1])
        differing = [u for i, u in enumerate(UTTS) if not np.array_equal(Y[u, :m], Y32[i, :m])]
        report = {'identical': len(UTTS) - len(differing), 'of': len(UTTS), 'ties': {}}
        W64 = O.cast_weights(W, np.float64) if differing else None
        eng.set_sharing(0, 0)
        for u in differing:
            Y64, tr64 = _f64_trace(ocfg, W64, raw[u])
            assert np.array_equal(Y64[0, :m], Y32[UTTS.index(u), :m]), 'the float64 oracle sides with the float32 one'
            s, ev = _departure(tg['tokens'], tg['parents'], tg['scores'], tr64, u * K)
            assert s is not None, 'hypotheses differ but no step does'
            _assert_tie(ev, ('batch', precision, u))
            report['ties'][u] = ev
            Xu = O.make_batch([raw[u]])
            Yu, tu = eng.decode_trace(Xu, 201, logits=False)
            if not np.array_equal(Yu[0, :m], Y32[UTTS.index(u), :m]):
                s1, ev1 = _departure(tu['tokens'], tu['parents'], tu['scores'], tr64, 0)
                assert s1 is not None
                _assert_tie(ev1, ('alone', precision, u))
                report['ties'][u]['alone'] = ev1
            else:
                report['ties'][u]['alone'] = 'identical to the oracle'
        print('near-tie report', precision, report)
        # whatever ties there are, they are rare: most of the sample is token-identical
        assert report['identical'] >= len(UTTS) - 3, report
    finally:
        eng.close()
