#!/usr/bin/env python3
"""Headline benchmark: decoded audio-seconds per second (RTFx) of the batched inference path
(frame stacking -> encoder -> beam-search decoder) on BASELINE.json's config 2:
char yaml architecture (hidden 1024, 16 heads, 6+6 blocks, GLU FFN 4096, tied V=3900), beam 10,
max_target_length 200, 64 synthetic 10-second utterances per GPU per step.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One process per GPU; every rank decodes its own 64-utterance batch (weak scaling, utterances are
independent), then the hypotheses are gathered with one RCCL all_gather of a fixed-width int32
buffer (the path's only exchange, model.py:192-201).  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md: FP32 matrix peak
PEAK_BF16_MFMA_TFLOPS = 2500.0   # same guide: BF16 dense MFMA peak
PEAK_HBM_GBS = 8000.0            # same guide: HBM3E 8 TB/s
# dense-layer arithmetic (include/asr_amd.h asr_precision): reported dtype and MFMA products per f32 multiply-add
PRECISION_INFO = {
    'f32': ('f32', 1, PEAK_F32_MFMA_TFLOPS),
    'f32_bf16x6': ('f32 (every f32 operand split exactly into 3 bf16 terms, 6 bf16 MFMA products per k-step, f32 accumulate: f32-chain accuracy)', 6,
                   PEAK_BF16_MFMA_TFLOPS),
    'f32_f16x3': ('f32 (every f32 operand split into 2 fp16 terms x1 + 2^-11 x2, 3 fp16 MFMA products per k-step in two f32 '
                  'accumulators: f32-chain accuracy)', 3, PEAK_BF16_MFMA_TFLOPS),
    'bf16x3': ('bf16x3 (2-term bf16 split, 3 MFMA products, f32 accumulate)', 3, PEAK_BF16_MFMA_TFLOPS),
    'bf16': ('bf16 (f32 accumulate; attention, layer norms and beam search in f32)', 1, PEAK_BF16_MFMA_TFLOPS),
}


def bench_config(beam=10, mtl=200, precision='f32_f16x3', vocab=3900):
    from asr_amd.utils import AttrDict
    return AttrDict({
        'model': 'Transformer', 'dst_vocab_size': vocab, 'hidden_units': 1024, 'scale_embedding': True,
        'tie_embedding_and_softmax': True, 'num_blocks': 6, 'num_heads': 16, 'ff_activation': 'glu',
        'train': {'input_dim': 320},
        'test': {'batch_size': 64, 'max_target_length': mtl, 'lp_alpha': 0.6, 'beam_size': beam, 'num_gpus': 1,
                 'precision': precision},
    })


def measured_traffic(kernel_class, precision):
    """HBM-side bytes per launch of `kernel_class` from the committed rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE,
    collected and corrected by tools/pmc_traffic.py on this same command); None when that class was not profiled."""
    best = None
    pdir = os.path.join(ROOT, 'profiles')
    for f in sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []:
        if f.endswith('_pmc_traffic.json') and (('_' + precision + '_') in f or (precision == 'f32' and f.count('_') == 2)):
            try:
                best = json.load(open(os.path.join(pdir, f))).get(kernel_class, best)
            except (OSError, ValueError):
                pass
    return None if best is None else best.get('traffic_bytes_per_launch')


def synthetic_raw(n, t_raw, seed):
    rng = np.random.default_rng(seed)
    return rng.standard_normal((n, t_raw, 80)).astype(np.float32)


def cpu_baseline(cfg, W, raw, sample_utts):
    """The oracle (CPU restatement of the reference algorithm, K/V-cache formulation: the variant most
    favourable to the CPU) timed on this box's host cores over a bounded sample of the same workload."""
    from oracle import asr_oracle as O
    try:
        import threadpoolctl
        info = threadpoolctl.threadpool_info()
        cores = max([i.get('num_threads', 1) for i in info] or [1])
    except Exception:
        cores = int(os.environ.get('OMP_NUM_THREADS', '1'))
    ocfg = O.make_config(hidden_units=cfg.hidden_units, num_heads=cfg.num_heads, num_blocks=cfg.num_blocks,
                         dst_vocab_size=cfg.dst_vocab_size, input_dim=cfg.train.input_dim,
                         ff_activation=cfg.ff_activation, scale_embedding=cfg.scale_embedding,
                         tie_embedding_and_softmax=cfg.tie_embedding_and_softmax, beam_size=cfg.test.beam_size,
                         max_target_length=cfg.test.max_target_length, lp_alpha=cfg.test.lp_alpha)
    X = O.make_batch([raw[i] for i in range(sample_utts)])
    t0 = time.time()
    Y = O.build_test_model(X, W, ocfg, num_gpus=1, algo='kv')
    dt = time.time() - t0
    audio_s = sample_utts * raw.shape[1] / 100.0
    return {'value': audio_s / dt, 'unit': 'audio-sec/sec', 'cores': int(cores), 'kind': 'port',
            'sample': '%d of the 64 utterances (%.0f audio-s), beam %d, %d decode steps, numpy/OpenBLAS float32, '
                      'K/V-cache formulation of the reference algorithm; %.1f s wall'
                      % (sample_utts, audio_s, cfg.test.beam_size, Y.shape[1], dt)}, Y


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--t-raw', type=int, default=1000)
    ap.add_argument('--beam', type=int, default=10)
    ap.add_argument('--max-target-length', type=int, default=200)
    ap.add_argument('--vocab', type=int, default=3900, help='dst_vocab_size: 3900 = char config (the headline), 1388 = pinyin, '
                    '28444 = word, 3205 = SEAME')
    ap.add_argument('--cpu-sample-utts', type=int, default=4)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--precision', default=os.environ.get('ASR_AMD_PRECISION', 'f32_f16x3'), choices=sorted(PRECISION_INFO))
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))

    import torch
    import torch.distributed as dist
    from asr_amd import weights
    from asr_amd.model import Engine

    # Rehearsal switch for a one-GPU box (never set by the driver): all ranks share GPU 0 and the gather runs over
    # gloo on host tensors, so the multi-rank plumbing can be exercised where RCCL cannot (two ranks on one device).
    rehearse = os.environ.get('ASR_BENCH_REHEARSE_ON_ONE_GPU', '') == '1'
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if rehearse:
            dist.init_process_group(backend='gloo')
        else:
            dist.init_process_group(backend='nccl', device_id=torch.device('cuda', local_rank))

    cfg = bench_config(args.beam, args.max_target_length, args.precision, args.vocab)
    W = weights.random_init(cfg, seed=0)
    eng = Engine(cfg, local_rank)
    eng.load_weights(W)

    n, t_raw = args.batch, args.t_raw
    raw = synthetic_raw(n, t_raw, seed=1000 + rank)
    d_raw = torch.from_numpy(raw).cuda(local_rank)
    torch.cuda.synchronize()
    width = args.max_target_length + 1
    front = (t_raw, 80, 3, 0, 3)

    def step():
        Y = eng.decode_dev(d_raw.data_ptr(), n, 0, 0, raw=front)
        if world > 1:      # gather the hypotheses (fixed width, padded with 3) on every rank
            buf = torch.full((n, width), 3, dtype=torch.int32)
            buf[:, :Y.shape[1]] = torch.from_numpy(Y)
            if not rehearse:
                buf = buf.cuda(local_rank)
            out = [torch.empty_like(buf) for _ in range(world)]
            dist.all_gather(out, buf)
        return Y

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    Y = None
    for _ in range(args.warmup):
        Y = step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        Y = step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device='cpu' if rehearse else 'cuda')
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    if rank == 0:
        audio_s = world * n * t_raw / 100.0 * args.steps
        out = {
            'metric': 'decoded audio-sec/sec (RTFx) whole node, char unit512 block6, beam=%d' % args.beam,
            'value': audio_s / elapsed, 'unit': 'audio-sec/sec', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': 1000.0 * elapsed / max(args.steps, 1), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': PRECISION_INFO[args.precision][0], 'data': 'synthetic',
            'config': {'workload': '%s (hidden 1024, 16 heads, 6+6 blocks, GLU 4096, tied V=%d), beam %d, '
                                   'max_target_length %d, %d x %.0f s synthetic fbank utterances per GPU per step '
                                   '(T=%d stacked frames), random-init weights, %d decode steps taken'
                                   % ('char config' if args.vocab == 3900 else 'char architecture with another vocabulary',
                                      args.vocab, args.beam, args.max_target_length, n, t_raw / 100.0, (t_raw + 2) // 3,
                                      Y.shape[1]),
                       'batch_per_gpu': n, 'beam': args.beam, 'parallelism': 'dp%d' % world, 'precision': args.precision,
                       # precision f32_f16x3: calls the library repeated on the bf16x6 kernels because an operand left
                       # fp16's range (0 = every timed step ran on the fp16x3 kernels)
                       'range_retries': eng.range_retries()},
        }
        # ---- roofline of the dominant kernel: HIP events around every launch of one extra pass ----
        eng.set_profile(True)
        eng.decode_dev(d_raw.data_ptr(), n, 0, 0, raw=front)
        prof = eng.get_profile()
        eng.set_profile(False)
        overhead = [p for p in prof if p['name'].startswith('(')]
        prof = [p for p in prof if not p['name'].startswith('(')]
        tot_ms = sum(p['ms'] for p in prof) or 1.0
        dom = max(prof, key=lambda p: p['ms'])
        ach = dom['flops'] / (dom['ms'] * 1e-3) / 1e12 if dom['flops'] > 0 else dom['bytes'] / (dom['ms'] * 1e-3) / 1e9
        is_mfma = dom['flops'] > 0 and dom['name'].startswith('gemm')
        _, nprod, mfma_peak = PRECISION_INFO[args.precision]
        peak = mfma_peak if is_mfma else PEAK_HBM_GBS
        out['roofline'] = {
            'kernel': dom['name'], 'bound': 'mfma' if is_mfma else 'hbm', 'achieved': ach,
            'peak': peak, 'unit': 'TFLOP/s' if is_mfma else 'GB/s', 'frac': ach / peak,
            'traffic': measured_traffic(dom['name'], args.precision),
            'launches': dom['launches'], 'avg_us': 1000.0 * dom['ms'] / max(dom['launches'], 1),
            'share_of_kernel_time': dom['ms'] / tot_ms,
            # elapsed time of an EMPTY HIP-event pair, already subtracted from every bracket (asr_set_profile)
            'event_pair_overhead_us': (1000.0 * overhead[0]['ms'] / max(overhead[0]['launches'], 1)) if overhead else None,
        }
        if is_mfma and nprod > 1:
            # `achieved` counts ALGORITHMIC flops (2MNK); the matrix pipe executes nprod bf16 products per f32 one
            out['roofline']['mfma_products_per_multiply'] = nprod
            out['roofline']['executed_frac'] = nprod * ach / peak
        out['kernels'] = [{'name': p['name'], 'ms': round(p['ms'], 3), 'launches': p['launches'],
                           'tflops': round(p['flops'] / (p['ms'] * 1e-3) / 1e12, 2) if p['ms'] > 0 else 0.0,
                           'gbs': round(p['bytes'] / (p['ms'] * 1e-3) / 1e9, 1) if p['ms'] > 0 else 0.0}
                          for p in sorted(prof, key=lambda p: -p['ms'])]
        if world == 1 and not args.no_cpu_baseline:
            cb, Yc = cpu_baseline(cfg, W, raw, min(args.cpu_sample_utts, n))
            m = min(Yc.shape[1], Y.shape[1])
            cb['hypotheses_match_gpu'] = bool(np.array_equal(Yc[:, :m], Y[:Yc.shape[0], :m]))
            out['cpu_baseline'] = cb
        print(json.dumps(out), flush=True)

    eng.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
