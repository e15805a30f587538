"""north_star asks for the reference's beam sequences on the fixed synthetic set.  On bench.py's own batch (64 x 10 s,
random-init char model, 201 steps) a float32 implementation cannot promise that literally: random-init posteriors are
nearly flat, beam scores reach -600 after 100 steps, and now and then two candidate totals of the pruning step
(model.py:284-301) lie closer together than ONE float32 ulp of the score - which of them survives then depends on the
order a float32 sum was taken in (tile shapes, K-slices), in TensorFlow as much as here.  This test is the collected form
of round 2's diagnostics (tests/diag_near_tie.py, profiles/r02_check_modes.json): the GPU decodes the WHOLE 64-utterance
batch with the kernel shapes of the headline (shared-GPU regime, one run of 640-row kernels) in the default mode and in
exact f32; for utterances 0-5 and 14 (4 and 14 are the two that differed in round 2) every hypothesis must either equal
the float32 oracle's, or the test PROVES that the difference is a tie and not a defect.  The arbiter is the float64
oracle's per-step trace of that utterance (the float32 oracle is itself only ONE of the float32 roundings - it, too, may
leave the float64 search at such a tie, and then the GPU may be the one that agrees with float64):
  * for the GPU's batch trace AND for the float32 oracle's trace, the first step at which the (token, parent) rows leave the
    float64 oracle's is found; at least one of the two must leave it (their hypotheses differ);
  * at such a step the float64 oracle's own candidate totals around the pruning boundary are less than 2 float32 ulps of the
    score apart (a defect would show as a departure at a step with a clear gap), and up to that step the departing side's
    beam scores follow the float64 ones to float32 summation accuracy;
  * decoded alone (other tile shapes, other summation order) the utterance again either equals one of the two oracles or
    leaves the float64 search at a proven tie."""
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


def _oracle_trace(ocfg, Wd, raw_u, dtype):
    X = O.make_batch([raw_u]).astype(dtype)
    enc, _ = O.encoder(X, Wd, ocfg)
    tr = {}
    Y = O.beam_search(enc, Wd, ocfg, algo='kv', trace=tr)
    return Y, tr


def _departure(tokens, parents, scores, tr64, row0):
    """First step at which one utterance's K rows of a float32 trace (rows row0 .. row0+K; parents are global row indices)
    differ from the float64 oracle's single-utterance trace.  -> (step or None, evidence)."""
    tokens, parents, scores = np.asarray(tokens), np.asarray(parents), np.asarray(scores)
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
            i = UTTS.index(u)
            Y64, tr64 = _oracle_trace(ocfg, W64, raw[u], np.float64)
            Y32u, tr32 = _oracle_trace(ocfg, W, raw[u], np.float32)
            assert np.array_equal(Y32u[0, :m], Y32[i, :m])               # (the batch of seven decodes u as it decodes alone)
            s_gpu, ev_gpu = _departure(tg['tokens'], tg['parents'], tg['scores'], tr64, u * K)
            s_o32, ev_o32 = _departure(tr32['tokens'], tr32['parents'], tr32['scores'], tr64, 0)
            assert s_gpu is not None or s_o32 is not None, 'hypotheses differ but neither trace leaves the float64 search'
            row = {}
            if s_gpu is not None:
                _assert_tie(ev_gpu, ('gpu batch', precision, u))
                row['gpu_leaves_float64_at'] = ev_gpu
            else:
                assert np.array_equal(Y[u, :m], Y64[0, :m])
                row['gpu_leaves_float64_at'] = 'nowhere: the GPU has the float64 hypothesis'
            if s_o32 is not None:
                _assert_tie(ev_o32, ('float32 oracle', precision, u))
                row['float32_oracle_leaves_float64_at'] = ev_o32
            Yu, tu = eng.decode_trace(O.make_batch([raw[u]]), 201, logits=False)
            if np.array_equal(Yu[0, :m], Y32[i, :m]) or np.array_equal(Yu[0, :m], Y64[0, :m]):
                row['alone'] = 'identical to the float%d oracle' % (32 if np.array_equal(Yu[0, :m], Y32[i, :m]) else 64)
            else:
                s1, ev1 = _departure(tu['tokens'], tu['parents'], tu['scores'], tr64, 0)
                assert s1 is not None
                _assert_tie(ev1, ('gpu alone', precision, u))
                row['alone'] = ev1
            report['ties'][u] = row
        print('near-tie report', precision, report)
        # whatever ties there are, they are rare: most of the sample is token-identical
        assert report['identical'] >= len(UTTS) - 3, report
    finally:
        eng.close()
