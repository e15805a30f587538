#!/usr/bin/env python3
"""Headline benchmark: decoded audio-seconds per second (RTFx) of the batched inference path
(frame stacking -> encoder -> beam-search decoder) on BASELINE.json's config 2:
char yaml architecture (hidden 1024, 16 heads, 6+6 blocks, GLU FFN 4096, tied V=3900), beam 10,
max_target_length 200, 64 synthetic 10-second utterances per GPU per step.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One process per GPU; every rank decodes its own 64-utterance batch per step (weak scaling, utterances are
independent), then the step's hypotheses are gathered with one RCCL all_gather of a fixed-width int32
buffer (the path's only exchange, model.py:192-201).  Rank 0 prints ONE JSON line.
`python bench.py --gpus N` with N > 1 and no launcher (no RANK in the environment) starts the N ranks ITSELF: the parent
imports neither torch nor the HIP library, gives each child RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT,
relays rank 0's JSON line and exits non-zero when any rank fails.  Under an external launcher WORLD_SIZE must equal
--gpus (anything else is an error, never a silent one-GPU run).  `--dry-launch` rehearses only that plumbing: every
child joins a gloo group, the ranks are all-gathered, rank 0 prints {"dry_launch": true, "n_gpus": N, "ranks_seen": [...]}.
A step = one 64-utterance batch through the whole path.  By default FOUR steps are in flight on every GPU
(`--in-flight 4`: four engine handles, asr_amd.model.EnginePool, each decoding whole steps one after the other): the
K timed steps are the same K batches decoded the same way, `ms_per_step` = elapsed / K; the line also carries
`one_step_at_a_time` (the same measurement with `--in-flight 1`).  A handle decodes TWO steps per library call
(`--steps-per-decode 2`, what `Model.predict_batches` does by default with whole batches queued: every weight plane is
streamed once for both batches; each batch keeps its own hypotheses, bit for bit); `one_step_per_decode` is the same
measurement with every 64-utterance batch a call of its own.

The same line also carries, at N = 1 (all timed by this run, on this box):
  roofline            dominant kernel class, HIP events around every launch of one extra pass that launches the same
                      shapes as the timed passes (same half batches)
  cpu_baseline        the oracle's K/V-cache decode on a few utterances (the formulation most favourable to the CPU)
  cpu_baseline_ref_shaped   the oracle's LITERAL reference algorithm (model.py:559-633: block-output cache, K/V of the
                      history and of the beam-tiled encoder re-projected every step) on 1 utterance for 20 steps,
                      with the extrapolation to 201 steps by the algorithm's FLOP count stated
  alt_precisions      the other dense-layer arithmetics (f32_bf16x6, f32, bf16), a few steps each, with a flag saying
                      whether their hypotheses equal the headline mode's
  configs             every BASELINE.json configuration timed by this same run (a few steps each): config 1 (8 ragged
                      utterances, greedy; batch by batch and as one joint decode), 3 / 4 (512 ragged utterances through the
                      10-frame buckets, V = 1388 / 28444, batch by batch and with joint decodes of 64), 5 (50 x 30 s, V = 3205,
                      bf16 and the default mode), each with value, ms per step and a crc32 of its hypotheses

`--workload bucketed` instead runs BASELINE configs 3/4 as stated: ragged utterances, 10-frame length buckets
(utils.py:285-323), batches dealt over the ranks by cost, each rank decoding its share with the real engine, one
all_gather of the hypotheses; a step = the whole utterance set.
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

# Streams of one process share its hardware queues (4 by default) in creation order, and kernels of two streams that share
# a queue do not overlap.  The benchmark keeps several engine handles busy on one GPU, each on a stream of its own, so it
# asks the HIP runtime for 8 queues - before anything initialises it - by the policy the product path applies as well
# (asr_amd/hwqueues.py, called by asr_amd.model.Model / EnginePool): a lower value is raised, ASR_AMD_KEEP_HW_QUEUES=1 keeps
# what the environment says.  (Measured, 64-utterance steps: 3 handles on 4 queues 225 ms a step, 4 handles on 8 queues
# 214 ms; 4 handles on 4 queues 243 ms.)
from asr_amd import hwqueues  # noqa: E402  (pure Python: loads neither torch nor the HIP library)

HWQ = hwqueues.ensure()

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
    'bf16': ('bf16 (f32 accumulate; bf16 K/V caches, q and P as two bf16 terms; soft-max, layer norms and beam search in f32)', 1, PEAK_BF16_MFMA_TFLOPS),
}
MFMA_KERNELS = ('gemm', 'encoder_attention')          # classes priced against the matrix pipe; everything else against HBM


def bench_config(beam=10, mtl=200, precision='f32_f16x3', vocab=3900):
    from asr_amd.utils import AttrDict
    return AttrDict({
        'model': 'Transformer', 'dst_vocab_size': vocab, 'hidden_units': 1024, 'scale_embedding': True,
        'tie_embedding_and_softmax': True, 'num_blocks': 6, 'num_heads': 16, 'ff_activation': 'glu',
        'train': {'input_dim': 320},
        'test': {'batch_size': 64, 'max_target_length': mtl, 'lp_alpha': 0.6, 'beam_size': beam, 'num_gpus': 1,
                 'precision': precision},
    })


PMC_STEPS_PER_DECODE = 2          # the committed round-3 PMC passes ran the default command: 1280-row launches


def measured_traffic(kernel_class, precision, steps_per_decode=PMC_STEPS_PER_DECODE):
    """HBM-side bytes per launch of `kernel_class` from the committed rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE,
    collected and corrected by tools/pmc_traffic.py on this same command); None when that class was not profiled, or when
    this run launches other row counts than the profiled command did.  Later files (sorted by name: r03 after r02) win."""
    if steps_per_decode != PMC_STEPS_PER_DECODE:
        return None
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


def _oracle_config(cfg, mtl=None):
    from oracle import asr_oracle as O
    return O.make_config(hidden_units=cfg.hidden_units, num_heads=cfg.num_heads, num_blocks=cfg.num_blocks,
                         dst_vocab_size=cfg.dst_vocab_size, input_dim=cfg.train.input_dim,
                         ff_activation=cfg.ff_activation, scale_embedding=cfg.scale_embedding,
                         tie_embedding_and_softmax=cfg.tie_embedding_and_softmax, beam_size=cfg.test.beam_size,
                         max_target_length=cfg.test.max_target_length if mtl is None else mtl, lp_alpha=cfg.test.lp_alpha)


def _blas_threads():
    try:
        import threadpoolctl
        info = threadpoolctl.threadpool_info()
        return int(max([i.get('num_threads', 1) for i in info] or [1]))
    except Exception:
        return int(os.environ.get('OMP_NUM_THREADS', '1'))


def cpu_baseline(cfg, W, raw, sample_utts):
    """The oracle (CPU restatement of the reference algorithm, K/V-cache formulation: the variant most
    favourable to the CPU) timed on this box's host cores over a bounded sample of the same workload."""
    from oracle import asr_oracle as O
    ocfg = _oracle_config(cfg)
    X = O.make_batch([raw[i] for i in range(sample_utts)])
    t0 = time.time()
    Y = O.build_test_model(X, W, ocfg, num_gpus=1, algo='kv')
    dt = time.time() - t0
    audio_s = sample_utts * raw.shape[1] / 100.0
    return {'value': audio_s / dt, 'unit': 'audio-sec/sec', 'cores': _blas_threads(), 'kind': 'port',
            'sample': '%d of the %d utterances (%.0f audio-s), beam %d, %d decode steps, numpy/OpenBLAS float32, '
                      'K/V-cache formulation of the reference algorithm; %.1f s wall'
                      % (sample_utts, raw.shape[0], audio_s, cfg.test.beam_size, Y.shape[1], dt)}, Y


def decode_flops(cfg, t, steps, ref_shaped):
    """Algorithmic FLOPs of `steps` decode steps of ONE utterance at beam K (1 MAC = 2 FLOP).
    ref_shaped: the reference's own algorithm (self qkv over all i positions, cross K/V of the beam-tiled encoder
    re-projected every step, model.py:559-633); else the K/V-cache algorithm."""
    H, L, V, K = float(cfg.hidden_units), cfg.num_blocks, float(cfg.dst_vocab_size), cfg.test.beam_size
    tot = 0.0
    for i in range(1, steps + 1):
        if ref_shaped:
            per = L * (6 * H * H * i + 4 * H * H * t + (2 + 2 + 2 + 16 + 8) * H * H + 4 * H * (i + t)) + 2 * V * H
        else:
            per = L * (36 * H * H + 4 * H * (i + t)) + 2 * V * H
        tot += K * per
    return tot


def cpu_baseline_ref_shaped(cfg, W, raw, steps=20):
    """The LITERAL reference algorithm (oracle algo='ref') on ONE utterance for `steps` decode steps."""
    from oracle import asr_oracle as O
    ocfg = _oracle_config(cfg, mtl=steps - 1)               # loop runs min(T + 50, mtl) + 1 steps
    X = O.make_batch([raw[0]])
    t0 = time.time()
    enc, _ = O.encoder(X, W, ocfg)
    t_enc = time.time() - t0
    t0 = time.time()
    Y = O.beam_search(enc, W, ocfg, algo='ref')
    t_dec = time.time() - t0
    t = X.shape[1]
    full_steps = min(t + 50, cfg.test.max_target_length) + 1
    f_meas = decode_flops(cfg, t, Y.shape[1], True)
    f_full = decode_flops(cfg, t, full_steps, True)
    audio_s = raw.shape[1] / 100.0
    t_full = t_enc + t_dec * f_full / f_meas
    return {'value': audio_s / t_full, 'unit': 'audio-sec/sec', 'cores': _blas_threads(), 'kind': 'port',
            'measured': {'utterances': 1, 'decode_steps': int(Y.shape[1]), 'encoder_s': round(t_enc, 3), 'decode_s': round(t_dec, 3),
                         'decode_gflop': round(f_meas / 1e9, 1)},
            'extrapolation': 'value = audio seconds / (encoder_s + decode_s x %.1f): the %d measured steps scaled to the '
                             '%d steps the headline decode takes by the reference algorithm\'s own FLOP count '
                             '(%.0f GFLOP for %d steps vs %.0f GFLOP measured; the K/V-cache algorithm needs %.0f GFLOP '
                             'for the same %d steps)' % (f_full / f_meas, Y.shape[1], full_steps, f_full / 1e9, full_steps,
                                                          f_meas / 1e9, decode_flops(cfg, t, full_steps, False) / 1e9, full_steps),
            'sample': '1 of the utterances (%.0f audio-s), beam %d, %d decode steps of the literal model.py:559-633 algorithm '
                      '(block-output cache, per-step re-projection), numpy/OpenBLAS float32' % (audio_s, cfg.test.beam_size, Y.shape[1])}


def identical_to_oracle(Y, Yc):
    """How many of the oracle's sample utterances (the first rows of the batch, decoded by the float32 CPU oracle as a batch
    of their own - with equal-length utterances the padding, and with it Q1's statistics, are the same) the GPU reproduces
    token for token.  Random-init weights over 201 steps leave near-ties (candidate totals closer than one float32 ulp of
    the score): tests/test_gpu_near_tie.py shows, per differing utterance, that this is the only way they differ."""
    if Yc is None:
        return 0
    m = min(Yc.shape[1], Y.shape[1])
    return int(np.all(Yc[:, :m] == Y[:Yc.shape[0], :m], axis=1).sum())


def roofline_of(prof, precision, steps_per_decode=PMC_STEPS_PER_DECODE):
    overhead = [p for p in prof if p['name'].startswith('(')]
    prof = [p for p in prof if not p['name'].startswith('(')]
    tot_ms = sum(p['ms'] for p in prof) or 1.0
    dom = max(prof, key=lambda p: p['ms'])
    is_mfma = dom['flops'] > 0 and dom['name'].startswith(MFMA_KERNELS)
    _, nprod, mfma_peak = PRECISION_INFO[precision]
    if dom['name'].startswith('encoder_attention'):
        nprod, mfma_peak = 1, PEAK_F32_MFMA_TFLOPS                    # f32 MFMA in every mode
    if is_mfma:                                                        # matrix-bound: algorithmic FLOPs / time vs the MFMA peak
        ach, peak, unit = dom['flops'] / (dom['ms'] * 1e-3) / 1e12, mfma_peak, 'TFLOP/s'
    else:                                                              # HBM-bound: algorithmic BYTES / time vs the HBM peak
        ach, peak, unit = dom['bytes'] / (dom['ms'] * 1e-3) / 1e9, PEAK_HBM_GBS, 'GB/s'
    r = {
        'kernel': dom['name'], 'bound': 'mfma' if is_mfma else 'hbm', 'achieved': ach,
        'peak': peak, 'unit': unit, 'frac': ach / peak,
        'traffic': measured_traffic(dom['name'], precision, steps_per_decode),
        'launches': dom['launches'], 'avg_us': 1000.0 * dom['ms'] / max(dom['launches'], 1),
        'share_of_kernel_time': dom['ms'] / tot_ms,
        # elapsed time of an EMPTY HIP-event pair, already subtracted from every bracket (asr_set_profile)
        'event_pair_overhead_us': (1000.0 * overhead[0]['ms'] / max(overhead[0]['launches'], 1)) if overhead else None,
    }
    if is_mfma and nprod > 1:
        # `achieved` counts ALGORITHMIC flops (2MNK); the matrix pipe executes nprod 16-bit products per f32 one
        r['mfma_products_per_multiply'] = nprod
        r['executed_frac'] = nprod * ach / peak
    kernels = [{'name': p['name'], 'ms': round(p['ms'], 3), 'launches': p['launches'],
                'avg_us': round(1000.0 * p['ms'] / max(p['launches'], 1), 2),
                'tflops': round(p['flops'] / (p['ms'] * 1e-3) / 1e12, 2) if p['ms'] > 0 else 0.0,
                'gbs': round(p['bytes'] / (p['ms'] * 1e-3) / 1e9, 1) if p['ms'] > 0 else 0.0}
               for p in sorted(prof, key=lambda p: -p['ms'])]
    return r, kernels


def run_bucketed(args, rank, local_rank, world, rehearse, W=None):
    """BASELINE configs 1 / 3 / 4: ragged utterances, length-bucketed batches sharded over the ranks (asr_amd/dist.py).
    Returns the result line (a dict) on rank 0, None elsewhere."""
    import threading
    import torch
    import torch.distributed as dist
    from asr_amd import dist as D
    from asr_amd import weights
    from asr_amd.model import EnginePool

    cfg = bench_config(args.beam, args.max_target_length, args.precision, args.vocab)
    if W is None:
        W = weights.random_init(cfg, seed=0)
    pool = EnginePool(cfg, local_rank, max(1, args.engines_per_gpu))
    pool.load_weights(W)
    engines = pool.engines
    eng = engines[0]
    n_total = world * args.utts_per_gpu
    rng = np.random.default_rng(4242)                                  # every rank builds the same utterance list
    raw_lens = [int(x) for x in rng.integers(args.t_raw_min, args.t_raw_max + 1, size=n_total)]
    # left3 / sub3 stacking by the library's own front-end kernel (fea.py:142-149), outside the timed region
    feats = [eng.stack_frames(rng.standard_normal((ln, 80)).astype(np.float32)) for ln in raw_lens]
    lengths = [len(f) for f in feats]
    batches = D.bucket_batches(lengths, args.tokens_per_batch)
    costs = [D.batch_cost(len(b), max(lengths[i] for i in b), args.beam, args.max_target_length) for b in batches]
    owners = D.assign_batches(costs, world)
    model_load = [sum(costs[i] for i in o) for o in owners]
    busy = [0.0]
    busy_lock = threading.Lock()

    def make_decode_fn(e):
        def decode_fn(X):
            t0 = time.perf_counter()
            Y = e.decode_groups(X) if args.merge_utts > 0 else e.decode(X)
            with busy_lock:
                busy[0] += time.perf_counter() - t0
            return Y
        return decode_fn
    decode_fns = [make_decode_fn(e) for e in engines]

    device = None if (rehearse or world == 1) else torch.device('cuda', local_rank)

    def step():
        with pool.together():
            return D.decode_sharded(decode_fns, feats, args.tokens_per_batch, args.beam, args.max_target_length, rank=rank,
                                    world_size=world, device=device, merge_utts=args.merge_utts)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    busy[0] = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        Y = step()
    fence()
    elapsed = time.perf_counter() - t0
    busy_all, per_rank = [busy[0]], [elapsed]
    if world > 1:
        bt = torch.tensor([busy[0], elapsed, float(rank)], dtype=torch.float64, device='cpu' if rehearse else 'cuda')
        parts = [torch.empty_like(bt) for _ in range(world)]
        dist.all_gather(parts, bt)
        busy_all = [float(p[0].item()) for p in parts]
        per_rank = [float(p[1].item()) for p in parts]
        ranks_seen = sorted(int(p[2].item()) for p in parts)
        elapsed = max(per_rank)
    else:
        ranks_seen = [0]
    out = None
    if rank == 0:
        audio_s = sum(raw_lens) / 100.0 * args.steps
        out = {
            'metric': 'decoded audio-sec/sec (RTFx) whole node, length-bucketed ragged batches, beam=%d' % args.beam,
            'value': audio_s / elapsed, 'unit': 'audio-sec/sec', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': 1000.0 * elapsed / max(args.steps, 1), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': PRECISION_INFO[args.precision][0], 'data': 'synthetic',
            'config': {'workload': 'bucketed: %d utterances per GPU, T_raw ~ U[%d,%d] (%.0f audio-s in all), 10-frame length '
                                   'buckets closed above %d stacked frames (utils.py:285-323), %d batches dealt over %d rank(s) by '
                                   'cost and decoded by %d engine handle(s) per GPU side by side, host features (H2D inside the step), one all_gather of the hypotheses; char '
                                   'architecture with V=%d, beam %d, max_target_length %d, random-init weights'
                                   % (args.utts_per_gpu, args.t_raw_min, args.t_raw_max, sum(raw_lens) / 100.0, args.tokens_per_batch, len(batches), world,
                                      len(engines), args.vocab, args.beam, args.max_target_length),
                       'parallelism': 'dp%d' % world, 'precision': args.precision, 'batches': len(batches), 'engines_per_gpu': len(engines),
                       # > 0: a rank's batches joined into joint decodes of up to that many utterances (Engine.decode_groups)
                       'merge_utts': args.merge_utts,
                       'joint_decodes': [len(D.merge_jobs(o, [len(b) for b in batches], [max(lengths[i] for i in b) for b in batches],
                                                         args.merge_utts)) for o in owners],
                       'batch_sizes': sorted(len(b) for b in batches)[::max(1, len(batches) // 16)],
                       'hw_queues': HWQ['value'], 'hw_queues_source': HWQ['source'],
                       'range_retries': eng.range_retries()},
            # per-rank seconds inside Engine.decode over the timed steps (summed over the rank's engine threads), and how even the deal was
            'per_rank_busy_s': [round(b, 4) for b in busy_all],
            'per_rank_ms_per_step': [round(1000.0 * e / max(args.steps, 1), 3) for e in per_rank],
            'ranks_seen': ranks_seen,
            'load_balance': {'measured_max_over_mean': max(busy_all) / (sum(busy_all) / len(busy_all)),
                             'cost_model_max_over_mean': max(model_load) / (sum(model_load) / len(model_load))},
            'hypotheses_shape': list(Y.shape),
            'hypotheses_crc32': int(__import__('zlib').crc32(np.ascontiguousarray(Y).tobytes())),   # same for any number of engines / ranks
        }
    pool.close()
    return out


def plan_calls(k, handles, per_call):
    """k steps as library calls of at most `per_call` steps for a pool of `handles`: every handle gets the same number of steps
    (+- 1), in calls of `per_call` steps (a handle's odd last step makes its last call one larger), the large calls first - so
    that the last round of a finite run does not leave handles idle while one of them still decodes a pair (a stream of batches
    without end has no such tail)."""
    per_handle = [k // handles + (1 if i < k % handles else 0) for i in range(handles)]
    calls = []
    for steps in per_handle:
        q, r = divmod(steps, per_call)
        mine = [per_call] * q
        if r == 1 and q and per_call >= 2:
            mine[-1] += 1                        # a handle's odd last step joins its last call (one more batch in the same joint decode:
        elif r:                                  # same hypotheses, and no round of single-step calls at the end of the run)
            mine.append(r)
        calls += mine
    return sorted(calls, reverse=True)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1, help='ranks = GPUs of this node.  Without a launcher (no RANK in the environment) and N > 1 '
                    'this process starts the N ranks itself; under torch.distributed.run WORLD_SIZE must equal it')
    ap.add_argument('--steps', type=int, default=8)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--t-raw', type=int, default=1000)
    ap.add_argument('--beam', type=int, default=10)
    ap.add_argument('--max-target-length', type=int, default=200)
    ap.add_argument('--vocab', type=int, default=3900, help='dst_vocab_size: 3900 = char config (the headline), 1388 = pinyin, '
                    '28444 = word, 3205 = SEAME')
    ap.add_argument('--cpu-sample-utts', type=int, default=4)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-alt-precisions', action='store_true')
    ap.add_argument('--no-one-at-a-time', action='store_true', help='skip the one_step_at_a_time leg (profiling runs)')
    ap.add_argument('--no-configs', action='store_true', help='skip the `configs` leg (the other BASELINE configurations)')
    ap.add_argument('--alt-steps', type=int, default=8)
    ap.add_argument('--in-flight', type=int, default=4, help='steps (whole batches) being decoded at any time on one GPU, each by an '
                    'engine handle of its own (asr_amd.model.EnginePool); 1 = one step at a time')
    ap.add_argument('--steps-per-decode', type=int, default=2, choices=[1, 2, 3, 4], help='steps (64-utterance batches) a handle decodes per library '
                    'call: 2 = what Model.predict_batches does by default with whole batches queued (pairs of batches as one joint decode of '
                    '1280-row kernels: every weight plane is streamed once for both; every batch keeps its own padding, loop limit and '
                    'hypotheses), 1 = one batch per call')
    ap.add_argument('--precision', default=os.environ.get('ASR_AMD_PRECISION', 'f32_f16x3'), choices=sorted(PRECISION_INFO))
    ap.add_argument('--workload', default='fixed', choices=['fixed', 'bucketed'])
    ap.add_argument('--no-two-steps', action='store_true', help='skip the one-step-per-decode leg (--no-one-at-a-time skips it as well)')
    ap.add_argument('--merge-utts', type=int, default=0, help='--workload bucketed: join a rank\'s batches into joint decodes of up to this '
                    'many utterances (Engine.decode_groups); 0 = batch by batch like the reference\'s translate loop')
    ap.add_argument('--utts-per-gpu', type=int, default=512, help='--workload bucketed: utterances per GPU (a test set of a few hundred utterances fills the 10-frame buckets; 64 leaves mostly single-utterance batches)')
    ap.add_argument('--t-raw-min', type=int, default=200, help='--workload bucketed: raw frames per utterance ~ U[min, max]')
    ap.add_argument('--t-raw-max', type=int, default=1500)
    ap.add_argument('--engines-per-gpu', type=int, default=4, help='--workload bucketed: engine handles (K/V cache, '
                    'streams, one host thread each; one shared copy of the weights) that decode the rank\'s batches side by side')
    ap.add_argument('--tokens-per-batch', type=int, default=7000, help='--workload bucketed: stacked frames per batch '
                    '(test.tokens_per_batch of the SEAME yaml)')
    ap.add_argument('--dry-launch', action='store_true', help='rehearse the multi-rank plumbing only: every rank joins a gloo group and '
                    'reports its rank (no GPU, no HIP library)')
    return ap.parse_args(argv)


def launch_ranks(args, argv):
    """`--gpus N` with no launcher: start the N ranks, one process each.  This parent never imports torch and never touches
    HIP (a process that has initialised the GPU must not be the one that spawns or re-execs others): it only spawns, relays
    rank 0's stdout (the JSON line) and waits.  Any rank that exits non-zero ends the job: the others are terminated (by PID)
    and the parent exits with that code."""
    import signal
    import socket
    import subprocess
    import threading
    assert 'torch' not in sys.modules, 'the launching parent must not have imported torch'
    n = args.gpus
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({'RANK': str(r), 'LOCAL_RANK': str(r), 'WORLD_SIZE': str(n), 'LOCAL_WORLD_SIZE': str(n),
                    'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': str(port)})
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')       # dmabuf IPC: what RCCL needs on this driver
        procs.append(subprocess.Popen([sys.executable, '-u', os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0)))
    lines = []

    def relay():
        for line in procs[0].stdout:
            lines.append(line)
            sys.stdout.write(line)
            sys.stdout.flush()
    th = threading.Thread(target=relay, daemon=True)
    th.start()

    def stop_all(code):
        for p in procs:
            if p.poll() is None:
                p.terminate()
        deadline = time.time() + 10.0
        for p in procs:
            try:
                p.wait(timeout=max(0.1, deadline - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
        return code
    signal.signal(signal.SIGTERM, lambda *_: sys.exit(stop_all(143)))
    try:
        while True:
            codes = [p.poll() for p in procs]
            bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                print('bench.py: rank %d exited with code %d; stopping the other ranks' % bad[0], file=sys.stderr, flush=True)
                return stop_all(bad[0][1] if bad[0][1] > 0 else 1)
            if all(c == 0 for c in codes):
                break
            time.sleep(0.2)
    except KeyboardInterrupt:
        return stop_all(130)
    th.join(timeout=10.0)
    got = None
    for line in lines:
        try:
            d = json.loads(line)
            if isinstance(d, dict) and 'n_gpus' in d:
                got = d
        except ValueError:
            pass
    if got is None or got.get('n_gpus') != n:
        print('bench.py: rank 0 printed no result line with n_gpus == %d' % n, file=sys.stderr, flush=True)
        return 1
    return 0


def dry_rank(args, rank, world):
    """--dry-launch: the rank joins a gloo group, the ranks are all-gathered, rank 0 prints what it saw."""
    if os.environ.get('ASR_BENCH_DRY_FAIL_RANK', '') == str(rank):      # test hook: a rank that dies before it joins
        sys.exit(3)
    import torch
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    if world > 1:
        dist.init_process_group(backend='gloo')
        t = torch.tensor([rank], dtype=torch.int64)
        parts = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(parts, t)
        seen = sorted(int(p.item()) for p in parts)
        dist.barrier()
        dist.destroy_process_group()
    else:
        seen = [0]
    if rank == 0:
        print(json.dumps({'dry_launch': True, 'n_gpus': world, 'ranks_seen': seen, 'local_rank': int(os.environ.get('LOCAL_RANK', '0')),
                          'master_port': os.environ.get('MASTER_PORT')}), flush=True)


def _ns(args, **kw):
    d = dict(vars(args))
    d.update(kw)
    return argparse.Namespace(**d)


def measure_configs(args, local_rank, W_char):
    """Every BASELINE.json configuration timed by this same run, a few steps each (the headline, config 2, is `value`).
    Hypothesis crc32s: the same for any number of handles / ranks (every batch is decoded as alone); joint decodes change
    the float32 summation order of the row-wise kernels."""
    import zlib
    import torch
    from asr_amd import weights
    from asr_amd.model import EnginePool
    out = {}

    # config 1: the reference's CPU-runnable case - 8 ragged utterances, greedy
    c1 = dict(vocab=3900, beam=1, utts_per_gpu=8, t_raw_min=300, t_raw_max=1000, steps=3)
    t0 = time.time()

    def bucketed_w(name, what, W, **kw):
        a = _ns(args, workload='bucketed', steps=kw.pop('steps', 2), warmup=1, engines_per_gpu=4, **kw)
        r = run_bucketed(a, 0, local_rank, 1, False, W=W)
        out[name] = {'what': what, 'value': r['value'], 'unit': r['unit'], 'ms_per_step': r['ms_per_step'], 'steps': a.steps,
                     'precision': a.precision, 'batches': r['config']['batches'], 'merge_utts': a.merge_utts,
                     'hypotheses_crc32': r['hypotheses_crc32']}
    what1 = 'config 1: char V=3900, greedy (beam 1), 8 ragged utterances T_raw ~ U[300,1000], 10-frame buckets'
    bucketed_w('config1_batch_by_batch', what1 + ', batch by batch on four handles', W_char, merge_utts=0, **c1)
    bucketed_w('config1_merge_utts_8', what1 + ', one joint decode (--merge-utts 8)', W_char, merge_utts=8, **c1)
    for name, vocab in (('config3_pinyin', 1388), ('config4_word', 28444)):
        Wv = weights.random_init(bench_config(args.beam, args.max_target_length, args.precision, vocab), seed=0)
        what = ('%s: V=%d, beam %d, 512 ragged utterances T_raw ~ U[200,1500] through the 10-frame buckets (46 batches), four handles'
                % (name.replace('_', ' '), vocab, args.beam))
        bucketed_w(name + '_batch_by_batch', what + ', batch by batch', Wv, vocab=vocab, merge_utts=0, utts_per_gpu=512)
        bucketed_w(name + '_merge_utts_64', what + ', joint decodes of up to 64 utterances', Wv, vocab=vocab, merge_utts=64, utts_per_gpu=512)
        del Wv
    # config 5: SEAME vocabulary, 50 x 30 s; as in the headline, four handles each decode two steps (two 50-utterance batches)
    # per library call - 500 beam rows count as a whole batch (engine.hip WHOLE_BATCH_ROWS_DEFAULT), so a pair reproduces its
    # halves bit for bit, which the first utterances' crc32 of a step decoded alone checks here
    n5, t5, spd5 = 50, 3000, 2
    raw5 = np.concatenate([synthetic_raw(n5, t5, seed=5000 + i) for i in range(spd5)])
    d5 = torch.from_numpy(raw5).cuda(local_rank)
    W5 = weights.random_init(bench_config(args.beam, args.max_target_length, 'bf16', 3205), seed=0)
    for prec in ('bf16', args.precision):
        p5 = EnginePool(bench_config(args.beam, args.max_target_length, prec, 3205), local_rank, 4)
        p5.load_weights(W5)

        def dec5(e, nsteps):
            return e.decode_dev(d5.data_ptr(), nsteps * n5, 0, 0, raw=(t5, 80, 3, 0, 3))
        for _ in p5.map(dec5, [spd5] * 4):
            pass
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        k5 = 4 * spd5
        for Y5 in p5.map(dec5, [spd5] * 4):
            pass
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        Y5a = list(p5.map(dec5, [1]))[0]                               # the first batch alone (in the pool's regime)
        m5 = min(Y5a.shape[1], Y5.shape[1])
        out['config5_seame_30s_' + prec] = {
            'what': 'config 5: SEAME V=3205, 50 x 30 s utterances (T=1000 stacked frames), beam %d, %d decode steps, four handles x %d steps (50-utterance batches) per call'
                    % (args.beam, Y5.shape[1], spd5),
            'value': n5 * t5 / 100.0 * k5 / dt, 'unit': 'audio-sec/sec', 'ms_per_step': 1000.0 * dt / k5, 'steps': k5, 'precision': prec,
            'hypotheses_crc32': int(zlib.crc32(np.ascontiguousarray(Y5[:n5]).tobytes())),
            'utterances_identical_to_the_step_decoded_alone': '%d of %d' % (int(np.all(Y5a[:, :m5] == Y5[:n5, :m5], axis=1).sum()), n5)}
        p5.close()
    out['seconds_spent'] = round(time.time() - t0, 1)
    return out


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    if args.gpus < 1:
        sys.exit('bench.py: --gpus must be >= 1')
    launched = 'RANK' in os.environ
    if not launched and args.gpus > 1:
        sys.exit(launch_ranks(args, argv))                 # this process only spawns, relays and waits: no torch, no HIP
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        sys.exit('bench.py: --gpus %d but WORLD_SIZE=%d: the launcher must start exactly --gpus ranks (python -m torch.distributed.run '
                 '--nproc-per-node N ... bench.py --gpus N); without a launcher `python bench.py --gpus N` starts the ranks itself'
                 % (args.gpus, world))
    if args.dry_launch:
        dry_rank(args, rank, world)
        return

    import torch
    import torch.distributed as dist
    from asr_amd import weights

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

    if args.workload == 'bucketed':
        out = run_bucketed(args, rank, local_rank, world, rehearse)
        if out is not None:
            print(json.dumps(out), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    from asr_amd.model import EnginePool
    cfg = bench_config(args.beam, args.max_target_length, args.precision, args.vocab)
    W = weights.random_init(cfg, seed=0)
    # `--in-flight` engine handles on the GPU (asr_amd.model.EnginePool: own K/V cache and stream, one host thread each, ONE
    # shared copy of the weights): that many steps (whole 64-utterance batches) are being decoded at any time, each exactly
    # as if alone
    pool = EnginePool(cfg, local_rank, args.in_flight)
    pool.load_weights(W)
    eng = pool.engines[0]

    n, t_raw = args.batch, args.t_raw
    spd = args.steps_per_decode
    raw = synthetic_raw(n, t_raw, seed=1000 + rank)
    # the second batch of a pair is another batch (other utterances): nothing is shared between the two but the kernels
    # the largest call of the timed region (plan_calls: a handle's odd last step joins its last call) decides how many batches the
    # device buffer holds and what the warm-up calls look like
    largest = spd + 1 if spd >= 2 else 1
    raw_pair = np.concatenate([raw] + [synthetic_raw(n, t_raw, seed=2000 * i + rank) for i in range(1, largest)]) if largest > 1 else raw
    d_raw = torch.from_numpy(raw_pair).cuda(local_rank)
    torch.cuda.synchronize()
    width = args.max_target_length + 1
    front = (t_raw, 80, 3, 0, 3)
    seen_ranks = [[0]]

    def decode(e, _=None):                                   # one step: the first batch alone
        return e.decode_dev(d_raw.data_ptr(), n, 0, 0, raw=front)

    def decode_call(e, nsteps):                              # nsteps (1 .. largest) steps in one library call -> their hypotheses, step by step
        if nsteps > largest:
            raise ValueError('a call of %d steps, the device buffer holds %d batches' % (nsteps, largest))
        Y = e.decode_dev(d_raw.data_ptr(), nsteps * n, 0, 0, raw=front)
        return [Y[i * n:(i + 1) * n] for i in range(nsteps)]

    def gather(Y):          # the hypotheses of a step (fixed width, padded with 3; column 0 = the rank) on every rank: the path's only exchange
        if world > 1:
            buf = torch.full((n, width + 1), 3, dtype=torch.int32)
            buf[:, 0] = rank
            buf[:, 1:1 + Y.shape[1]] = torch.from_numpy(np.ascontiguousarray(Y))
            if not rehearse:
                buf = buf.cuda(local_rank)
            out = [torch.empty_like(buf) for _ in range(world)]
            dist.all_gather(out, buf)
            seen_ranks[0] = sorted(set(int(o[0, 0].item()) for o in out))

    def run_steps(p, k, per_call=None):
        """k steps through pool p, `per_call` of them per library call: results come back in step order on this thread, which
        gathers each step's hypotheses while the pool's threads go on decoding the next ones.  Returns the FIRST batch's Y."""
        per_call = per_call or spd
        calls = plan_calls(k, len(p.engines), per_call)
        Y = None
        for Ys in p.map(decode_call, calls):
            for Yi in Ys:
                gather(Yi)
            Y = Ys[0]
        return Y

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(p, k, per_call=None):
        fence()
        t0 = time.perf_counter()
        Y = run_steps(p, k, per_call)
        fence()
        return time.perf_counter() - t0, Y

    def warm(p, k_timed, per_call=None, at_least=0):
        """Untimed: one call of the largest size the timed plan for `k_timed` steps holds on every handle (a handle's caches grow
        with the largest call it has seen - that must not happen inside the timed region), then what is left of `at_least` steps."""
        per_call = per_call or spd
        big = max(plan_calls(k_timed, len(p.engines), per_call) + [1])
        Yw = None
        for Ys in p.map(decode_call, [big] * len(p.engines)):
            for Yi in Ys:
                gather(Yi)                        # (the collective is warmed as well)
            Yw = Ys[0]
        left = at_least - big * len(p.engines)
        if left > 0:
            Yw = run_steps(p, left, per_call)
        return Yw

    # W untimed steps - and at least one call of the timed region's largest size on every handle
    if args.warmup > 0:
        warm(pool, args.steps, spd, args.warmup)
    elapsed, Y = timed(pool, args.steps)
    per_rank = [elapsed]
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device='cpu' if rehearse else 'cuda')
        parts = [torch.empty_like(tt) for _ in range(world)]
        dist.all_gather(parts, tt)
        per_rank = [float(p.item()) for p in parts]
        elapsed = max(per_rank)                              # the job is done when its slowest rank is

    if rank == 0:
        audio_s = world * n * t_raw / 100.0 * args.steps
        nfl = len(pool.engines)
        out = {
            'metric': 'decoded audio-sec/sec (RTFx) whole node, char unit512 block6, beam=%d' % args.beam,
            'value': audio_s / elapsed, 'unit': 'audio-sec/sec', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': 1000.0 * elapsed / max(args.steps, 1), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': PRECISION_INFO[args.precision][0], 'data': 'synthetic',
            'config': {'workload': '%s (hidden 1024, 16 heads, 6+6 blocks, GLU 4096, tied V=%d), beam %d, '
                                   'max_target_length %d, %d x %.0f s synthetic fbank utterances per GPU per step '
                                   '(T=%d stacked frames), random-init weights, %d decode steps taken; %s'
                                   % ('char config' if args.vocab == 3900 else 'char architecture with another vocabulary',
                                      args.vocab, args.beam, args.max_target_length, n, t_raw / 100.0, (t_raw + 2) // 3,
                                      Y.shape[1],
                                      ('%d steps in flight per GPU: %d engine handles, each decoding %s as one run of %d-row kernels; '
                                       'ms_per_step = elapsed / steps'
                                       % (nfl * spd, nfl, 'whole steps one after the other' if spd == 1 else
                                          '%d steps (%d 64-utterance batches) per call (one more when a handle is left with an odd last step)' % (spd, spd), spd * n * args.beam))
                                      if nfl > 1 else 'one step at a time'),
                       'batch_per_gpu': n, 'beam': args.beam, 'parallelism': 'dp%d' % world, 'precision': args.precision,
                       'steps_in_flight': nfl * spd, 'engine_handles': nfl, 'steps_per_decode': spd, 'hw_queues': HWQ['value'], 'hw_queues_source': HWQ['source'],
                       # precision f32_f16x3: calls the library repeated on the bf16x6 kernels because an operand left
                       # fp16's range (0 = every timed step ran on the fp16x3 kernels)
                       'range_retries': sum(e.range_retries() for e in pool.engines),
                       'device_bytes_of_the_pool': pool.device_bytes()},
            # every rank's own clock over the timed steps, and the ranks whose rows arrived in the last step's all_gather
            'per_rank_ms_per_step': [round(1000.0 * e / max(args.steps, 1), 3) for e in per_rank],
            'ranks_seen': seen_ranks[0],
        }
        # ---- roofline of the dominant kernel: HIP events around every launch of one extra step on the first handle, on the
        # stream each kernel is launched on, left in the queues and read back afterwards (profile mode 2), while the
        # other handle(s) decode a step of their own beside it: the conditions of the timed region, i.e. the durations
        # rocprofv3 --kernel-trace reports for this command.  A second step with every bracket waited for and nothing
        # beside it (mode 1) gives each kernel alone on the GPU: `solo_us` ----
        import threading
        with pool.together():                     # the regime of the timed region (EnginePool.map enters it by itself)
            eng.set_profile(2)
            others = [threading.Thread(target=lambda e=e: [decode_call(e, spd) for _ in range(2)]) for e in pool.engines[1:]]
            for th in others:
                th.start()
            decode_call(eng, spd)                 # the call of the timed region: `spd` steps as one run of kernels
            prof = eng.get_profile()
            for th in others:
                th.join()
            eng.set_profile(1)
            decode_call(eng, spd)
            solo = {p['name']: p for p in eng.get_profile()}
            eng.set_profile(0)
        out['roofline'], out['kernels'] = roofline_of(prof, args.precision, spd)
        for k in out['kernels']:
            p1 = solo.get(k['name'])
            if p1 and p1['launches']:
                k['solo_us'] = round(1000.0 * p1['ms'] / p1['launches'], 2)
        # `achieved` / `frac` are per launch AS TIMED, i.e. with the other handle's kernels sharing the GPU (a kernel then
        # has about half the chip); the same class alone on the GPU:
        rf = out['roofline']
        rf['steps_in_flight'] = nfl * spd
        rf['steps_per_launch'] = spd               # a launch of the profiled call serves `spd` steps (64-utterance batches)
        p1 = solo.get(rf['kernel'])
        if p1 and p1['ms'] > 0:
            rf['solo_avg_us'] = 1000.0 * p1['ms'] / p1['launches']
            rf['solo_frac'] = rf['frac'] * rf['avg_us'] / rf['solo_avg_us']
        # The step as a whole (all kernel classes of ONE 64-utterance step, over the time a step takes in the timed
        # region, i.e. with the other steps in flight sharing the chip): what the chip sustains, as opposed to what one
        # kernel gets of it.  executed = algorithmic x the 16-bit products per multiply of the dense layers' kernels.
        nprod_mode = PRECISION_INFO[args.precision][1]
        step_s = 1e-3 * out['ms_per_step']                  # per GPU: every rank decodes a batch of its own per step
        alg = sum(p['flops'] for p in prof if not p['name'].startswith('(')) / spd          # (the profiled call decoded `spd` steps)
        exe = sum(p['flops'] * (nprod_mode if p['name'].startswith('gemm') else 1) for p in prof if not p['name'].startswith('(')) / spd
        ws = {'algorithmic_tflop': alg / 1e12, 'algorithmic_tflops': alg / step_s / 1e12,
              'executed_tflops': exe / step_s / 1e12, 'executed_frac_of_mfma_peak': exe / step_s / 1e12 / PRECISION_INFO[args.precision][2],
              'algorithmic_gb': sum(p['bytes'] for p in prof if not p['name'].startswith('(')) / spd / 1e9}
        ws['algorithmic_gbs'] = ws['algorithmic_gb'] / step_s
        ws['algorithmic_frac_of_hbm_peak'] = ws['algorithmic_gbs'] / PEAK_HBM_GBS
        tr = [(measured_traffic(p['name'], args.precision, spd) or measured_traffic(p['name'] + '_reg', args.precision, spd), p['launches'])
              for p in prof if not p['name'].startswith('(')]
        if all(t is not None for t, _ in tr):
            # memory-side traffic (PMC FETCH_SIZE / WRITE_SIZE per launch, profiles/): reads that missed the L2s, served by
            # the Infinity Cache or by HBM
            ws['pmc_traffic_gb'] = sum(t * l for t, l in tr) / spd / 1e9
            ws['pmc_traffic_gbs'] = ws['pmc_traffic_gb'] / step_s
            ws['pmc_traffic_frac_of_hbm_peak'] = ws['pmc_traffic_gbs'] / PEAK_HBM_GBS
        rf['whole_step'] = {k: round(v, 4) for k, v in ws.items()}
        Yc = None
        if world == 1 and not args.no_cpu_baseline:
            cb, Yc = cpu_baseline(cfg, W, raw, min(args.cpu_sample_utts, n))
            cb['hypotheses_match_gpu'] = bool(identical_to_oracle(Y, Yc) == Yc.shape[0])
            cb['utterances_identical_to_oracle'] = '%d of %d' % (identical_to_oracle(Y, Yc), Yc.shape[0])
            out['cpu_baseline'] = cb
            out['cpu_baseline_ref_shaped'] = cpu_baseline_ref_shaped(cfg, W, raw)
        pool.close()
        pool = None
        if world == 1 and nfl > 1 and not args.no_one_at_a_time:
            # the same steps one at a time (one handle, the library's default: two half batches on two streams)
            p1 = EnginePool(cfg, local_rank, 1)
            p1.load_weights(W)
            warm(p1, args.alt_steps, 1)
            dt, Y1 = timed(p1, args.alt_steps, 1)
            m = min(Y1.shape[1], Y.shape[1])
            same = np.all(Y1[:, :m] == Y[:, :m], axis=1)
            out['one_step_at_a_time'] = {'value': n * t_raw / 100.0 * args.alt_steps / dt, 'unit': 'audio-sec/sec',
                                         'ms_per_step': 1000.0 * dt / max(args.alt_steps, 1), 'steps': args.alt_steps,
                                         'utterances_identical': '%d of %d' % (int(same.sum()), n)}
            p1.close()
        if world == 1 and nfl > 1 and spd > 1 and not (args.no_two_steps or args.no_one_at_a_time):
            # one step per library call on every handle (round 2's headline setting): the same steps, every 64-utterance
            # batch a call of its own
            p2 = EnginePool(cfg, local_rank, nfl)
            p2.load_weights(W)
            warm(p2, args.alt_steps, 1)
            dt, Y2 = timed(p2, args.alt_steps, 1)
            m = min(Y2.shape[1], Y.shape[1])
            same = np.all(Y2[:, :m] == Y[:, :m], axis=1)
            out['one_step_per_decode'] = {'value': n * t_raw / 100.0 * args.alt_steps / dt, 'unit': 'audio-sec/sec',
                                          'ms_per_step': 1000.0 * dt / max(args.alt_steps, 1), 'steps': args.alt_steps, 'engine_handles': nfl,
                                          'utterances_identical': '%d of %d' % (int(same.sum()), n)}
            p2.close()
        if world == 1 and not args.no_alt_precisions:
            # the other dense-layer arithmetics on the same batch, timed by this same run the same way (fresh handles;
            # the headline's are closed first)
            alts = {}
            for prec in ('f32_bf16x6', 'f32', 'bf16'):
                if prec == args.precision:
                    continue
                p2 = EnginePool(bench_config(args.beam, args.max_target_length, prec, args.vocab), local_rank, args.in_flight)
                p2.load_weights(W)
                warm(p2, args.alt_steps)                                 # one call of the timed size on every handle
                dt, Ya = timed(p2, args.alt_steps)
                m = min(Ya.shape[1], Y.shape[1])
                same = np.all(Ya[:, :m] == Y[:, :m], axis=1)
                alts[prec] = {'value': n * t_raw / 100.0 * args.alt_steps / dt, 'unit': 'audio-sec/sec',
                              'ms_per_step': 1000.0 * dt / max(args.alt_steps, 1),
                              'steps': args.alt_steps, 'dtype': PRECISION_INFO[prec][0].split(' (')[0],
                              # against the default mode's hypotheses on this batch: random-init weights over 201 steps
                              # leave near-ties, so one f32-rounding-level difference can turn an utterance's beam
                              'hypotheses_match': bool(Ya.shape == Y.shape and same.all()),
                              'utterances_identical': '%d of %d' % (int(same.sum()), n)}
                if Yc is not None:
                    alts[prec]['utterances_identical_to_oracle'] = '%d of %d' % (identical_to_oracle(Ya, Yc), Yc.shape[0])
                p2.close()
            out['alt_precisions'] = alts
        if world == 1 and not args.no_configs and args.vocab == 3900 and args.precision == 'f32_f16x3':
            out['configs'] = measure_configs(args, local_rank, W)
        print(json.dumps(out), flush=True)

    if pool is not None:
        pool.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
