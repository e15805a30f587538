"""Multi-GPU data parallelism for the inference path: one process per GPU (torch.distributed;
backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests).

The reference's only cross-device step is the pad-with-3 + concat of per-tower predictions
(model.py:192-201); utterances are independent, so length-bucketed batches are dealt to ranks by
cost, each rank decodes its batches with its own weight replica and K/V cache, and ONE all_gather of a
fixed-width int32 buffer per evaluation restores the original utterance order.  No tensor parallelism.
"""
from __future__ import annotations

import os

import numpy as np


def world():
    """(world_size, rank) of this process: the initialised torch.distributed group, else the launcher's environment
    (`python -m torch.distributed.run`, `bench.py --gpus N`: WORLD_SIZE / RANK), else (1, 0)."""
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_world_size(), dist.get_rank()
    except ImportError:
        pass
    return int(os.environ.get('WORLD_SIZE', '1') or 1), int(os.environ.get('RANK', '0') or 0)


def init_from_env():
    """One process per GPU: join the process group the launcher described (MASTER_ADDR / MASTER_PORT / RANK / WORLD_SIZE).
    Backend "nccl" (= RCCL over xGMI) with this rank's GPU (LOCAL_RANK) as its device; ASR_AMD_DIST_BACKEND=gloo keeps the
    exchange on host tensors (CPU tests, several ranks rehearsed on one GPU).  Returns the torch device gathered buffers
    must live on (None = host).  No-op (None) for a single process."""
    n, _ = world()
    if n <= 1:
        return None
    import torch
    import torch.distributed as dist
    backend = os.environ.get('ASR_AMD_DIST_BACKEND', 'nccl')
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    if backend == 'gloo':
        if not dist.is_initialized():
            dist.init_process_group(backend='gloo')
        return None
    dev = torch.device('cuda', int(os.environ.get('LOCAL_RANK', '0') or 0))
    torch.cuda.set_device(dev)
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, device_id=dev)
    return dev


def bucket_batches(lengths, tokens_per_batch, bucket_min=50, bucket_max=10000, bucket_step=10):
    """Indices of utterances grouped like DataReader.get_test_batches_with_buckets
    (utils.py:285-323): 10-frame buckets on the stacked length, a bucket is closed once its frame
    count exceeds `tokens_per_batch`, leftovers in bucket order.  Returns a list of index lists."""
    buckets = list(range(bucket_min, bucket_max, bucket_step))
    caches = {}
    out = []
    for i, sl in enumerate(lengths):
        b = next((x for x in buckets if sl < x), None)
        if b is None:
            raise ValueError('The sequence is too long: (%d)' % sl)
        c = caches.setdefault(b, [[], 0])
        c[0].append(i)
        c[1] += int(sl)
        if c[1] > tokens_per_batch:
            out.append(c[0])
            caches[b] = [[], 0]
    for b in buckets:
        if b in caches and caches[b][0]:
            out.append(caches[b][0])
    return out


def batch_cost(n, t, beam, max_target_length, hidden=1024, blocks=6):
    """Rough FLOP cost of one batch: encoder + cross K/V + worst-case decode."""
    steps = min(t + 50, max_target_length) + 1
    enc = n * blocks * (32.0 * t * hidden * hidden + 4.0 * t * t * hidden) + n * blocks * 4.0 * t * hidden * hidden
    dec = n * beam * steps * blocks * (36.0 * hidden * hidden + 4.0 * hidden * (steps / 2 + t))
    return enc + dec


def assign_batches(costs, world_size):
    """Longest-processing-time-first deal of batches to ranks; returns rank -> list of batch indices."""
    order = sorted(range(len(costs)), key=lambda i: -costs[i])
    load = [0.0] * world_size
    owner = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda x: (load[x], x))
        owner[r].append(i)
        load[r] += costs[i]
    for r in range(world_size):
        owner[r].sort()
    return owner


def gather_hypotheses(local_ids, local_hyps, n_total, width, group=None, device=None):
    """all_gather of (utterance index, hypothesis) rows -> int32 [n_total, width] in original order on
    every rank.  `local_hyps` is a list of 1-D int32 arrays (already cut or padded with 3)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    cap = torch.tensor([len(local_ids)], dtype=torch.int64, device=device)
    if world > 1:
        dist.all_reduce(cap, op=dist.ReduceOp.MAX, group=group)
    cap = int(cap.item())
    buf = np.full((cap, width + 1), 3, np.int32)
    buf[:, 0] = -1
    for row, (i, h) in enumerate(zip(local_ids, local_hyps)):
        buf[row, 0] = i
        h = np.asarray(h, np.int32)[:width]
        buf[row, 1:1 + len(h)] = h
    t = torch.from_numpy(buf)
    if device is not None:
        t = t.to(device)
    if world > 1:
        parts = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(parts, t, group=group)
        allrows = torch.cat(parts, dim=0).cpu().numpy()
    else:
        allrows = t.cpu().numpy()
    out = np.full((n_total, width), 3, np.int32)
    seen = np.zeros(n_total, bool)
    for row in allrows:
        if row[0] >= 0:
            out[row[0]] = row[1:]
            seen[row[0]] = True
    if not seen.all():
        raise RuntimeError('hypotheses missing for utterances %s' % np.nonzero(~seen)[0][:8].tolist())
    return out


def merge_jobs(batch_ids, sizes, tmax, merge_utts):
    """Jobs for joint decodes (`Engine.decode_groups`): the batches of `batch_ids` in order of padded length, joined while
    a job stays within `merge_utts` utterances (neighbours in length share most of their loop: similar limits, similar
    key counts).  Returns a list of lists of batch ids; merge_utts <= 0: one batch per job."""
    if merge_utts <= 0:
        return [[b] for b in batch_ids]
    jobs, cur, n = [], [], 0
    for b in sorted(batch_ids, key=lambda b: (tmax[b], b)):
        if cur and n + sizes[b] > merge_utts:
            jobs.append(cur)
            cur, n = [], 0
        cur.append(b)
        n += sizes[b]
    if cur:
        jobs.append(cur)
    return jobs


def decode_sharded(decode_fn, feats, tokens_per_batch, beam, max_target_length, rank=0, world_size=1,
                   group=None, device=None, merge_utts=0):
    """Bucket `feats` (list of [T_u, D] arrays) by length, deal the batches over ranks, decode this
    rank's batches with `decode_fn(X) -> Y`, gather.  Returns int32 [len(feats), max_target_length+1].

    `decode_fn` may be a list of callables (one per engine handle on this rank's GPU, each with its own
    weight replica, K/V cache and streams): the rank's batches are then decoded by that many host threads
    side by side, most expensive first.  Bucketed batches are small (ten-odd utterances) and a small batch is
    bound by kernel latency, not by the GPU, so two or three of them share the GPU almost for free; the
    hypotheses do not depend on it (utterances are independent, every batch is decoded exactly as alone).

    merge_utts > 0: the rank's batches are joined into joint decodes of up to that many utterances (`merge_jobs`) and
    `decode_fn` is called with a LIST of padded batches and returns the list of their Y (`Engine.decode_groups`: every batch
    keeps its own padded length everywhere it enters the arithmetic; one beam search over all of them)."""
    fns = list(decode_fn) if isinstance(decode_fn, (list, tuple)) else [decode_fn]
    lengths = [len(f) for f in feats]
    batches = bucket_batches(lengths, tokens_per_batch)
    tmaxes = [max(lengths[i] for i in b) for b in batches]
    costs = [batch_cost(len(b), tmaxes[bi], beam, max_target_length) for bi, b in enumerate(batches)]
    mine = assign_batches(costs, world_size)[rank]

    def make(bi):
        idx = batches[bi]
        tmax = max(lengths[i] for i in idx)
        X = np.zeros((len(idx), tmax, feats[idx[0]].shape[1]), np.float32)
        for r, i in enumerate(idx):
            X[r, :lengths[i]] = feats[i]
        return X

    results = {}
    jobs = merge_jobs(mine, [len(b) for b in batches], tmaxes, merge_utts)

    def run(fn, job):
        if merge_utts > 0:
            Ys = fn([make(bi) for bi in job])
            if len(Ys) != len(job):
                raise RuntimeError('decode_fn returned %d results for %d batches' % (len(Ys), len(job)))
            for bi, Y in zip(job, Ys):
                results[bi] = Y
        else:
            results[job[0]] = fn(make(job[0]))

    if len(fns) == 1 or len(jobs) < 2:
        for job in jobs:
            run(fns[0], job)
    else:
        import queue
        import threading
        todo = queue.Queue()
        for job in sorted(jobs, key=lambda j: -sum(costs[b] for b in j)):
            todo.put(job)
        errors = []

        def worker(fn):
            while not errors:
                try:
                    job = todo.get_nowait()
                except queue.Empty:
                    return
                try:
                    run(fn, job)
                except Exception as ex:          # re-raised on the calling thread
                    errors.append(ex)
        threads = [threading.Thread(target=worker, args=(fn,)) for fn in fns]
        for th in threads:
            th.start()
        for th in threads:
            th.join()
        if errors:
            raise errors[0]
    ids, hyps = [], []
    for bi in mine:
        for r, i in enumerate(batches[bi]):
            ids.append(i)
            hyps.append(results[bi][r])
    return gather_hypotheses(ids, hyps, len(feats), max_target_length + 1, group=group, device=device)


def predict_batches_sharded(batches, predict_batches, beam, max_target_length, rank=0, world_size=1, group=None, device=None):
    """The multi-process form of `Model.predict_batches`: `batches` is the list of feature batches X [n_b, T_b, D] exactly as
    the single-process reader yields them (the SAME list on every rank: batch composition and padding decide the
    hypotheses, Q1 / Q7), whole batches are dealt to the ranks by cost, this rank's batches go through
    `predict_batches(iterable of X) -> iterable of Y` (so `test.engines_per_gpu` / `test.merge_utts` apply per rank), and ONE
    all_gather of a fixed-width int32 buffer brings every hypothesis to every rank (= the pad-with-3 + concat of
    model.py:192-201).  Entries of `batches` this rank does not own may be replaced by their shapes (n_b, T_b).
    Returns the list of Y per batch, int32 [n_b, max_target_length + 1] padded with 3, identical on every rank."""
    shapes = [(b[0], b[1]) if isinstance(b, tuple) else (b.shape[0], b.shape[1]) for b in batches]
    costs = [batch_cost(n, t, beam, max_target_length) for n, t in shapes]
    mine = assign_batches(costs, world_size)[rank] if world_size > 1 else list(range(len(batches)))
    first = np.concatenate([[0], np.cumsum([n for n, _ in shapes])]).astype(np.int64)
    for bi in mine:
        if isinstance(batches[bi], tuple):
            raise ValueError('batch %d belongs to rank %d but only its shape was kept' % (bi, rank))
    ids, hyps = [], []
    for bi, Y in zip(mine, predict_batches(batches[bi] for bi in mine)):
        if len(Y) != shapes[bi][0]:
            raise RuntimeError('predict_batches returned %d rows for a batch of %d' % (len(Y), shapes[bi][0]))
        for r in range(len(Y)):
            ids.append(int(first[bi]) + r)
            hyps.append(Y[r])
    if len(ids) != sum(shapes[bi][0] for bi in mine):
        raise RuntimeError('predict_batches ended early')
    allh = gather_hypotheses(ids, hyps, int(first[-1]), max_target_length + 1, group=group, device=device)
    return [allh[first[bi]:first[bi + 1]] for bi in range(len(batches))]
