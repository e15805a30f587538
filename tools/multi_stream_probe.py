"""Development probe: decode one 64-utterance batch as S concurrent sub-batches (S handles on one GPU, one host
thread each) and report throughput.  python tools/multi_stream_probe.py [streams ...]"""
import os, sys, time, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import bench_config, synthetic_raw
from asr_amd import weights
from asr_amd.model import Engine

cfg = bench_config(10, 200, os.environ.get('ASR_AMD_PRECISION', 'f32'))
W = weights.random_init(cfg, seed=0)
n, t_raw = 64, 1000
raw = synthetic_raw(n, t_raw, 1000)
d_raw = torch.from_numpy(raw).cuda(0)
torch.cuda.synchronize()
front = (t_raw, 80, 3, 0, 3)
ref = None
for S in [int(a) for a in sys.argv[1:]] or [1, 2, 4]:
    engs = [Engine(cfg, 0) for _ in range(S)]
    for e in engs:
        e.load_weights(W)
    per = n // S
    outs = [None] * S
    def work(i):
        ptr = d_raw.data_ptr() + i * per * t_raw * 80 * 4
        outs[i] = engs[i].decode_dev(ptr, per, 0, 0, raw=front)
    def step():
        th = [threading.Thread(target=work, args=(i,)) for i in range(S)]
        [t.start() for t in th]; [t.join() for t in th]
        L = max(o.shape[1] for o in outs)
        return np.concatenate([np.pad(o, ((0, 0), (0, L - o.shape[1])), constant_values=3) for o in outs])
    step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        Y = step()
    dt = (time.perf_counter() - t0) / 3
    if ref is None:
        ref = Y
    print('streams %d: %.1f ms per batch, %.0f audio-s/s, same hypotheses as 1 stream: %s' % (S, dt * 1e3, n * t_raw / 100 / dt, bool(ref.shape == Y.shape and np.array_equal(ref, Y))), flush=True)
    for e in engs:
        e.close()
