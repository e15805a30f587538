"""Experiment (GPU box): bench.py's 64-utterance batch decoded by N engine handles side by side (one host thread each,
every handle decoding whole batches one after the other), each with S sub-batches: batches per second of the whole GPU.

    python tools/inflight_probe.py  > gpurun_out/inflight.log
"""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def main():
    import torch
    from asr_amd import weights
    from asr_amd.model import Engine
    torch.cuda.set_device(0)
    combos = [(1, 2), (1, 1), (2, 1), (2, 2), (3, 1), (4, 1)]
    if len(sys.argv) > 1:
        combos = [tuple(int(x) for x in a.split('x')) for a in sys.argv[1:]]
    cfg0 = bench.bench_config(10, 200, 'f32_f16x3', 3900)
    W = weights.random_init(cfg0, seed=0)
    raw = bench.synthetic_raw(64, 1000, seed=1000)
    d_raw = torch.from_numpy(raw).cuda(0)
    torch.cuda.synchronize()
    front = (1000, 80, 3, 0, 3)
    ref = None
    for n_eng, sb in combos:
        cfg = bench.bench_config(10, 200, 'f32_f16x3', 3900)
        cfg['test']['sub_batches'] = sb
        engines = [Engine(cfg, 0) for _ in range(n_eng)]
        for e in engines:
            e.load_weights(W)
        per = 4
        outs = [None] * n_eng

        errors = []

        def work(j):
            try:
                for _ in range(per):
                    outs[j] = engines[j].decode_dev(d_raw.data_ptr(), 64, 0, 0, raw=front)
            except Exception as ex:
                errors.append(ex)
        for e in engines:                                     # warm-up
            e.decode_dev(d_raw.data_ptr(), 64, 0, 0, raw=front)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        th = [threading.Thread(target=work, args=(j,)) for j in range(n_eng)]
        for x in th:
            x.start()
        for x in th:
            x.join()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if errors:
            print('%d engine(s) x %d sub-batch(es): FAILED: %s' % (n_eng, sb, errors[0]), flush=True)
            sys.exit(1)
        nb = n_eng * per
        if ref is None:
            ref = outs[0]
        same = all(o.shape == ref.shape and np.array_equal(o, ref) for o in outs)
        print('%d engine(s) x %d sub-batch(es): %.1f ms per 64-utterance batch, %.0f audio-s/s, hypotheses identical: %s'
              % (n_eng, sb, 1000.0 * dt / nb, nb * 640.0 / dt, same), flush=True)
        for e in engines:
            e.close()


if __name__ == '__main__':
    main()
