"""Development probe (GPU box): Model.predict_batches - the path under Evaluator.translate - on 16 host batches of 64 x 10 s
utterances (stacked features on the host, H2D inside), for the test.engines_per_gpu / test.merge_utts settings of INTEGRATION.md.
    python tools/predict_batches_probe.py [batches [engines,merge ...]]"""
import os
import sys
import time
import zlib

os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import bench  # noqa: E402
from asr_amd import weights  # noqa: E402
from asr_amd.model import Engine, Model  # noqa: E402


def main():
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    settings = [(1, 0), (4, 0), (4, 128), (2, 128), (4, 256)] if len(sys.argv) < 3 else [tuple(int(x) for x in a.split(',')) for a in sys.argv[2:]]
    base = bench.bench_config(10, 200, 'f32_f16x3', 3900)
    W = weights.random_init(base, seed=0)
    e0 = Engine(base, 0)
    e0.load_weights(W)
    rng = np.random.default_rng(7)
    one = np.stack([e0.stack_frames(rng.standard_normal((1000, 80)).astype(np.float32)) for _ in range(64)])
    e0.close()
    batches = [np.ascontiguousarray(np.roll(one, i, axis=0)) for i in range(nb)]
    for engines, merge in settings:
        cfg = bench.bench_config(10, 200, 'f32_f16x3', 3900)
        cfg.test['engines_per_gpu'] = engines
        cfg.test['merge_utts'] = merge or None
        m = Model(cfg, 1, devices=[0])
        m.build_test_model()
        m.restore(W)
        list(m.predict_batches(batches[:max(engines, 2)]))                     # warm-up: every handle once
        t0 = time.perf_counter()
        Ys = list(m.predict_batches(batches))
        dt = time.perf_counter() - t0
        crc = 0
        for Y in Ys:
            Yp = np.full((64, 201), 3, np.int32)
            Yp[:, :Y.shape[1]] = Y
            crc = zlib.crc32(Yp.tobytes(), crc)
        print('engines_per_gpu %d merge_utts %3d: %7.1f audio-s/s  %6.1f ms per batch  crc32 %d'
              % (engines, merge, nb * 640.0 / dt, 1000.0 * dt / nb, crc), flush=True)
        m.close()


if __name__ == '__main__':
    main()
