"""Development probe (GPU box): Model.predict_batches - the path under Evaluator.translate - on host batches of 64 x 10 s
utterances (stacked features on the host, H2D inside), for the settings of INTEGRATION.md.  The first setting is the
UNMODIFIED configuration (no build-extension key, nothing in the environment): whatever it reaches is what a user of the
reference's YAML gets.
    python tools/predict_batches_probe.py [batches [default | engines,merge,pair ...]]
    python tools/predict_batches_probe.py shipped [batches [default | per_tower ...]]     (the reference YAML's own test section)"""
import os
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from asr_amd import weights  # noqa: E402
from asr_amd.model import Model  # noqa: E402
from asr_amd.utils import AttrDict  # noqa: E402


def config():
    return AttrDict({
        'model': 'Transformer', 'dst_vocab_size': 3900, 'hidden_units': 1024, 'scale_embedding': True,
        'tie_embedding_and_softmax': True, 'num_blocks': 6, 'num_heads': 16, 'ff_activation': 'glu',
        'train': {'input_dim': 320},
        'test': {'batch_size': 64, 'max_target_length': 200, 'lp_alpha': 0.6, 'beam_size': 10, 'num_gpus': 1},
    })


def shipped():
    """The test section the reference's char YAML ships (config_template_char_unit512_block6_left3_big_dim80_sp.yaml:49-54):
    batch_size 100, beam_size 13, num_gpus 4 - here in a process with ONE GPU: every batch is split into four towers' shards."""
    nb = int(sys.argv[2]) if len(sys.argv) > 2 else 24
    settings = sys.argv[3:] or ['default', 'per_tower']
    cfg0 = config()
    cfg0.test.update({'batch_size': 100, 'beam_size': 13, 'num_gpus': 4})
    W = weights.random_init(cfg0, seed=0)
    rng = np.random.default_rng(7)
    one = first = None
    for setting in settings:
        cfg = config()
        cfg.test.update({'batch_size': 100, 'beam_size': 13, 'num_gpus': 4})
        if setting == 'per_tower':                       # what the build did before: one handle, one call per tower, one after the other
            cfg.test['engines_per_gpu'] = 1
            cfg.test['joint_towers'] = False
        m = Model(cfg, 4, devices=[0])
        m.build_test_model()
        m.restore(W)
        if one is None:
            one = np.stack([m.engines[0].stack_frames(rng.standard_normal((1000, 80)).astype(np.float32)) for _ in range(100)])
        batches = [np.ascontiguousarray(np.roll(one, i, axis=0)) for i in range(nb)]
        list(m.predict_batches(batches[:len(m.pools[0].engines)]))
        t0 = time.perf_counter()
        Ys = list(m.predict_batches(batches))
        dt = time.perf_counter() - t0
        crc = 0
        for Y in Ys:
            Yp = np.full((100, 201), 3, np.int32)
            Yp[:, :Y.shape[1]] = Y
            crc = zlib.crc32(Yp.tobytes(), crc)
        flat = np.concatenate([np.pad(Y, ((0, 0), (0, 201 - Y.shape[1])), constant_values=3) for Y in Ys])
        same = '' if first is None else ', %d of %d utterances identical to the first setting' % (int(np.all(flat == first, axis=1).sum()), flat.shape[0])
        first = flat if first is None else first
        print('shipped test section (batch 100, beam 13, 4 towers on one GPU), %-9s (handles in the pool: %d): %7.1f audio-s/s  %6.1f ms per batch  crc32 %d%s'
              % (setting, len(m.pools[0].engines), nb * 1000.0 / dt, 1000.0 * dt / nb, crc, same), flush=True)
        m.close()


def main():
    if len(sys.argv) > 1 and sys.argv[1] == 'shipped':
        return shipped()
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 48
    settings = ['default', '1,0,0', '4,0,0', '4,0,1', '4,256,0'] if len(sys.argv) < 3 else sys.argv[2:]
    W = weights.random_init(config(), seed=0)
    rng = np.random.default_rng(7)
    one = None
    for setting in settings:
        cfg = config()
        if setting != 'default':
            engines, merge, pair = (int(x) for x in setting.split(','))
            cfg.test['engines_per_gpu'] = engines
            cfg.test['merge_utts'] = merge or None
            cfg.test['pair_batches'] = bool(pair)
        m = Model(cfg, 1, devices=[0])
        m.build_test_model()
        m.restore(W)
        if one is None:
            one = np.stack([m.engines[0].stack_frames(rng.standard_normal((1000, 80)).astype(np.float32)) for _ in range(64)])
        batches = [np.ascontiguousarray(np.roll(one, i, axis=0)) for i in range(nb)]
        list(m.predict_batches(batches[:2 * len(m.pools[0].engines)]))                     # warm-up: every handle once
        t0 = time.perf_counter()
        Ys = list(m.predict_batches(batches))
        dt = time.perf_counter() - t0
        crc = 0
        for Y in Ys:
            Yp = np.full((64, 201), 3, np.int32)
            Yp[:, :Y.shape[1]] = Y
            crc = zlib.crc32(Yp.tobytes(), crc)
        print('%-10s (engines_per_gpu, merge_utts, pair_batches; handles in the pool: %d, GPU_MAX_HW_QUEUES=%s): %7.1f audio-s/s  %6.1f ms per batch  crc32 %d'
              % (setting, len(m.pools[0].engines), os.environ.get('GPU_MAX_HW_QUEUES'), nb * 640.0 / dt, 1000.0 * dt / nb, crc), flush=True)
        m.close()


if __name__ == '__main__':
    main()
