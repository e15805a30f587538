"""(Test infrastructure, not collected by pytest: it imports the CPU oracle, which only tests/, bench.py's CPU leg and smoke() may.)
Diagnostic (GPU box): for single utterances of bench.py's batch, find the first decode step at which a precision
mode's (tokens, parents) leave the float64 oracle's, and print how close the oracle's own candidates are there: the gap
between neighbouring totals around the pruning boundary, against the error of the GPU's candidate scores on the steps
before.  A gap below that error is a near-tie (either answer is as good); a gap far above it is a defect.

    python tests/diag_near_tie.py 4 14 > gpurun_out/near_tie.json
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from oracle import asr_oracle as O  # noqa: E402


def main():
    utts = [int(a) for a in sys.argv[1:]] or [4, 14]
    from asr_amd import weights
    from asr_amd.model import Engine
    cfg = bench.bench_config(10, 200, 'f32_f16x3', 3900)
    ocfg = bench._oracle_config(cfg)
    K = ocfg.beam_size
    W = weights.random_init(cfg, seed=0)
    raw = bench.synthetic_raw(64, 1000, seed=1000)
    W64 = O.cast_weights(W, np.float64)
    out = {}
    for u in utts:
        X = O.make_batch([raw[u]])
        enc, _ = O.encoder(X.astype(np.float64), W64, ocfg)
        tr = {}
        Y64 = O.beam_search(enc, W64, ocfg, algo='kv', trace=tr)
        steps = Y64.shape[1]
        row = {'steps': steps}
        for prec in ('f32_f16x3', 'f32_bf16x6', 'f32'):
            e = Engine(bench.bench_config(10, 200, prec, 3900), 0)
            e.load_weights(W)
            Y, tg = e.decode_trace(X, steps)
            e.close()
            first = None
            for s in range(min(steps, len(tg['tokens']))):
                if not (np.array_equal(tg['tokens'][s], tr['tokens'][s]) and np.array_equal(tg['parents'][s], tr['parents'][s])):
                    first = s
                    break
            r = {'final_identical': bool(Y.shape == Y64.shape and np.array_equal(Y, Y64)), 'first_step_differing': first}
            if first is not None:
                # the oracle's K*K candidate totals at that step, from its own previous scores and logits
                prev = tr['scores'][first - 1] if first > 0 else np.array([0.0] + [-1e10] * (K - 1))
                z = O.log_softmax(tr['logits'][first])
                cs, ct = O.top_k(z, K)
                tot = np.sort((prev[:, None] + cs).reshape(-1))[::-1][:K + 3]
                r['oracle_totals_top'] = [float(x) for x in tot]
                r['gaps'] = [float(tot[i] - tot[i + 1]) for i in range(len(tot) - 1)]
                r['min_gap_in_top_K_plus_1'] = float(min(tot[i] - tot[i + 1] for i in range(K)))
                # error of the GPU's beam scores on the steps before (same beams, so directly comparable)
                errs = [float(np.max(np.abs(tg['scores'][s] - tr['scores'][s])[tr['scores'][s] > -1e9])) for s in range(first)]
                r['max_abs_score_err_before'] = max(errs) if errs else 0.0
                r['score_err_last_step_before'] = errs[-1] if errs else 0.0
                gs, os_ = tg['scores'][first], tr['scores'][first]
                r['gpu_scores_at_step'] = [float(x) for x in gs[:K]]
                r['oracle_scores_at_step'] = [float(x) for x in os_[:K]]
            row[prec] = r
            print(u, prec, r.get('first_step_differing'), r.get('min_gap_in_top_K_plus_1'), r.get('score_err_last_step_before'),
                  file=sys.stderr, flush=True)
        out[str(u)] = row
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
