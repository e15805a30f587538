"""Hardware queues of the process (HIP runtime variable GPU_MAX_HW_QUEUES; no counterpart in the reference, whose
TensorFlow session owns its streams).

The streams of one process share its hardware queues - four unless the variable says otherwise - in creation order, and
kernels of two streams that share a queue never overlap.  An `EnginePool` keeps one engine handle per batch in flight,
each on a stream of its own, so four handles need more than the default four queues (the fourth would share one with
the runtime's own stream).  Measured, 64-utterance batches: 3 handles on 4 queues 225 ms a batch, 4 on 8 queues 214 ms,
4 on 4 queues 243 ms.

The runtime reads the variable once, when the process first touches the GPU.  `ensure()` is therefore called by
everything that is about to create several handles (`bench.py` at import, `EnginePool`, `Model`) and applies ONE policy:
a value below what is wanted - unset, or set by the environment - is raised (and the change is logged and returned), a
value at or above it is kept, and `ASR_AMD_KEEP_HW_QUEUES=1` leaves whatever the environment says alone.  Once the GPU is
in use a change can no longer take effect: then nothing is changed, a warning is logged and `late` is reported, so that
callers can size their pools for the queues they really have."""
from __future__ import annotations

import logging
import os
import sys

DEFAULT_RUNTIME_QUEUES = 4          # HIP's own default
WANTED = 8                          # a queue per handle of a four-handle pool, with room for the runtime's and torch's streams
_gpu_started = [False]


def mark_gpu_started():
    """Called by asr_amd.model just before its first library call that initialises the HIP runtime."""
    _gpu_started[0] = True


def gpu_in_use():
    if _gpu_started[0]:
        return True
    torch = sys.modules.get('torch')
    try:
        return bool(torch is not None and torch.cuda.is_initialized())
    except Exception:
        return False


def current():
    try:
        v = int(os.environ.get('GPU_MAX_HW_QUEUES', '') or 0)
    except ValueError:
        v = 0
    return v if v > 0 else DEFAULT_RUNTIME_QUEUES


def ensure(wanted=WANTED):
    """-> {'value': queues the process has (or will have), 'source': how that came about, 'late': True when a raise
    was wanted but the GPU was already in use}."""
    raw = os.environ.get('GPU_MAX_HW_QUEUES', '')
    have = current()
    if have >= wanted:
        return {'value': have, 'source': 'environment (kept)', 'late': False}
    if os.environ.get('ASR_AMD_KEEP_HW_QUEUES', '') == '1':
        return {'value': have, 'source': 'environment (kept: ASR_AMD_KEEP_HW_QUEUES=1)', 'late': False}
    if gpu_in_use():
        logging.warning('asr_amd: GPU_MAX_HW_QUEUES=%s and the process already uses the GPU: it keeps %d hardware queues '
                        '(set GPU_MAX_HW_QUEUES=%d before the first GPU call to overlap four engine handles)',
                        raw or '(unset)', have, wanted)
        return {'value': have, 'source': 'too late to raise (GPU already in use)', 'late': True}
    os.environ['GPU_MAX_HW_QUEUES'] = str(wanted)
    src = 'raised from %s by asr_amd' % raw if raw else 'set by asr_amd (was unset)'
    logging.info('asr_amd: GPU_MAX_HW_QUEUES %s -> %d', raw or '(unset)', wanted)
    return {'value': wanted, 'source': src, 'late': False}
