"""Host-side mirror of the reference's `transformer/model.py` inference surface.

`Transformer(config, num_gpus)` + `build_test_model()` + `restore(...)` + `predict(X)` stand in for
the TF graph built by `Model.build_test_model` (model.py:162-204) and the
`sess.run(model.prediction, feed_dict=expand_feed_dict({model.src_pls: X}))` call of
`Evaluator.beam_search` (evaluate.py:46-47).  All arithmetic happens in libasr_amd.so (HIP, gfx950);
this file only shards, pads and concatenates like the reference's towers do (Q10).
"""
from __future__ import annotations

import ctypes as C
import logging
import os
import threading

import numpy as np

from . import _lib, hwqueues
from ._lib import lib
from .utils import shard_bounds

# Arithmetic of the dense layers when the YAML has no `test.precision` key and ASR_AMD_PRECISION is unset: float32
# operands split into two scaled fp16 terms (x = x1 + 2^-11 x2), three fp16 matrix products per k-step in two f32
# accumulators - measured against float64 its error is below that of the exact-f32 matrix instructions (same test bars,
# hypotheses identical) at 5.3x their rate.  Operands outside fp16's range make the library repeat the call on the bf16x6
# kernels (`f32_bf16x6`: three bf16 terms, six products, no range limit); `precision: f32` selects the exact-f32 matrix
# instructions (include/asr_amd.h asr_precision).
DEFAULT_PRECISION = 'f32_f16x3'
# beam rows from which a handle that shares the GPU picks its K-slices independently of the row count (engine.hip decode_splits,
# logit_split_for): batches of at least that many rows can be decoded in pairs with bit-identical hypotheses
PAIR_MIN_ROWS = 448                 # = engine.hip WHOLE_BATCH_ROWS_DEFAULT: from here on a shared GPU's K-slices do not follow the row count


def asr_config_from(config, device=0):
    """YAML config (AttrDict; the keys model.py:19-45,434-442 read) -> (`asr_config` struct of include/asr_amd.h,
    precision name).  Pure host code: no library call, so CPU tests can check what every reference YAML derives."""
    t = config.test
    c = _lib.asr_config()
    c.input_dim = int(config.train.input_dim)
    c.hidden_units = int(config.hidden_units)
    c.num_heads = int(config.num_heads)
    c.num_blocks = int(config.num_blocks)
    c.dst_vocab_size = int(config.dst_vocab_size)
    act = config.ff_activation or 'relu'
    if act not in _lib.FF_ACTIVATIONS:
        raise ValueError('unknown ff_activation %r' % act)
    c.ff_activation = _lib.FF_ACTIVATIONS[act]
    c.scale_embedding = 1 if config.scale_embedding else 0
    c.tie_embedding_and_softmax = 1 if config.tie_embedding_and_softmax else 0
    c.beam_size = int(t.beam_size)
    c.max_target_length = int(t.max_target_length)
    c.lp_alpha = float(t.lp_alpha)
    c.device = int(device)
    # build extension (not a reference key): arithmetic of the dense layers, see include/asr_amd.h asr_precision
    prec = (t.precision if t.precision is not None else os.environ.get('ASR_AMD_PRECISION')) or DEFAULT_PRECISION
    if prec not in _lib.PRECISIONS:
        raise ValueError('unknown test.precision %r (one of %s)' % (prec, sorted(_lib.PRECISIONS)))
    c.precision = _lib.PRECISIONS[prec]
    # build extension: 1 = one run of kernels per batch, 2 = two independent halves on two streams, None = library default
    sb = int(t.sub_batches or 0)
    if sb not in (0, 1, 2, 3, 4):
        raise ValueError('test.sub_batches must be 1..4')
    c.sub_batches = sb
    # build extension: other handles decode on the same GPU at the same time (EnginePool sets it for its handles)
    c.shared_gpu = 1 if t.shared_gpu else 0
    return c, prec


class Engine(object):
    """One libasr_amd handle = one GPU's weights, K/V caches and workspaces."""

    def __init__(self, config, device=0):
        c, self.precision = asr_config_from(config, device)
        self.cfg = c
        self.device = int(device)
        self._h = C.c_void_p()
        hwqueues.mark_gpu_started()
        rc = lib.asr_create(C.byref(c), C.byref(self._h))
        _lib.check(None, rc)
        self.hidden = c.hidden_units
        self.vocab = c.dst_vocab_size
        self.beam = c.beam_size
        self.input_dim = c.input_dim
        self.max_len = c.max_target_length + 1

    def close(self):
        if getattr(self, '_h', None):
            lib.asr_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- weights ---------------------------------------------------------------------------
    def load_weights(self, variables):
        """`variables`: mapping TF variable name -> array (names of SURVEY.md A.1; optimizer slots,
        `global_step`, `beta?_power` are skipped by the library)."""
        for name, arr in variables.items():
            a = np.asarray(arr)
            if a.dtype.kind not in 'fiu':
                continue
            if name.startswith('global_step') or a.dtype.kind in 'iu':
                continue
            a = _lib.f32(a)
            shape = (C.c_int64 * max(a.ndim, 1))(*a.shape)
            _lib.check(self._h, lib.asr_load_tensor(self._h, name.encode('utf-8'), _lib.fptr(a), shape, a.ndim))
        _lib.check(self._h, lib.asr_finalize_weights(self._h))

    def share_weights(self, other):
        """Use the packed device weights `other` (a finalized Engine of the same GPU, model shape and precision) holds,
        instead of load_weights: one copy per GPU for all handles of a pool (`asr_share_weights`)."""
        _lib.check(self._h, lib.asr_share_weights(self._h, other._h))

    def set_sharing(self, shared_gpu, sub_batches):
        """Regime of the calls that follow (`asr_set_sharing`): shared_gpu 1 = other handles decode on this GPU at the same
        time; sub_batches 0 = automatic, 1..4 = parts of a batch on their own streams."""
        _lib.check(self._h, lib.asr_set_sharing(self._h, int(shared_gpu), int(sub_batches)))

    # ---- the hot path ---------------------------------------------------------------------------
    def decode(self, X):
        """X float32 [n, t, input_dim] (zero rows = padding) -> int32 [n, len], padded with 3."""
        X = _lib.f32(X)
        n, t, d = X.shape
        if n == 0:
            return np.zeros((0, 0), np.int32)
        Y = np.empty((n, self.max_len), np.int32)
        ln = C.c_int32(0)
        _lib.check(self._h, lib.asr_decode(self._h, _lib.fptr(X), n, t, d, _lib.iptr(Y), Y.size, C.byref(ln)))
        return np.ascontiguousarray(Y.reshape(-1)[:n * ln.value].reshape(n, ln.value))

    def decode_groups(self, Xs):
        """Several batches, each float32 [n_g, t_g, input_dim] padded to its OWN t_g (the bucketed reader's batches), decoded
        as one beam search (`asr_decode_groups`: every batch through the encoder with its own T, the decode steps over all
        utterances together) -> list of int32 [n_g, len], padded with 3; an utterance's hypothesis up to its first 3 is
        what `decode` of its own batch gives (up to float32 summation order)."""
        Xs = [_lib.f32(X) for X in Xs if X.shape[0] > 0]
        if not Xs:
            return []
        d = Xs[0].shape[2]
        if any(X.ndim != 3 or X.shape[2] != d for X in Xs):
            raise ValueError('every batch must be [n, t, input_dim]')
        ng = np.array([X.shape[0] for X in Xs], np.int32)
        tg = np.array([X.shape[1] for X in Xs], np.int32)
        flat = np.concatenate([X.reshape(-1) for X in Xs])
        n = int(ng.sum())
        Y = np.empty((n, self.max_len), np.int32)
        ln = C.c_int32(0)
        _lib.check(self._h, lib.asr_decode_groups(self._h, _lib.fptr(flat), len(Xs), _lib.iptr(ng), _lib.iptr(tg), d, _lib.iptr(Y), Y.size,
                                                  C.byref(ln)))
        Y = Y.reshape(-1)[:n * ln.value].reshape(n, ln.value)
        out, r = [], 0
        for k in ng:
            out.append(np.ascontiguousarray(Y[r:r + k]))
            r += int(k)
        return out

    def loss(self, X, Y):
        """== sess.run(model.loss_sum) for one tower: X float32 [n, t, d], Y int32 [n, ty] (0 = padding) -> float."""
        X = _lib.f32(X)
        Y = _lib.i32(Y)
        n, t, d = X.shape
        if n == 0:
            return 0.0
        if Y.ndim != 2 or Y.shape[0] != n:
            raise ValueError('Y must be [n, ty] with the batch size of X')
        out = C.c_double(0.0)
        _lib.check(self._h, lib.asr_loss(self._h, _lib.fptr(X), n, t, d, _lib.iptr(Y), Y.shape[1], C.byref(out)))
        return float(out.value)

    def decode_dev(self, ptr, n, t, d, raw=None):
        """Device-resident input (bench): `ptr` = device address of X [n,t,d], or with
        raw=(raw_frames, raw_dim, left, right, sub) of RAW fbank [n, raw_frames, raw_dim]."""
        rf, rd, left, right, sub = raw if raw else (0, 0, 0, 0, 1)
        Y = np.empty((n, self.max_len), np.int32)
        ln = C.c_int32(0)
        _lib.check(self._h, lib.asr_decode_dev(self._h, C.c_void_p(ptr), n, t, d, rf, rd, left, right, sub,
                                               _lib.iptr(Y), Y.size, C.byref(ln)))
        return np.ascontiguousarray(Y.reshape(-1)[:n * ln.value].reshape(n, ln.value))

    def decode_trace(self, X, trace_steps, logits=True):
        """decode + the first `trace_steps` steps' pruned scores / tokens / parents (global row indices) and, unless
        logits=False, the pre-softmax logits [steps, rows, vocab] (asr_decode_trace; one run of kernels, no graph replay)."""
        X = _lib.f32(X)
        n, t, d = X.shape
        R = n * self.beam
        Y = np.empty((n, self.max_len), np.int32)
        ln = C.c_int32(0)
        logits = np.zeros((trace_steps, R, self.vocab), np.float32) if logits else None
        scores = np.zeros((trace_steps, R), np.float32)
        tokens = np.zeros((trace_steps, R), np.int32)
        parents = np.zeros((trace_steps, R), np.int32)
        _lib.check(self._h, lib.asr_decode_trace(self._h, _lib.fptr(X), n, t, d, _lib.iptr(Y), Y.size, C.byref(ln),
                                                 trace_steps, _lib.fptr(logits), _lib.fptr(scores),
                                                 _lib.iptr(tokens), _lib.iptr(parents)))
        Y = np.ascontiguousarray(Y.reshape(-1)[:n * ln.value].reshape(n, ln.value))
        return Y, dict(logits=logits, scores=scores, tokens=tokens, parents=parents)

    def encode(self, X):
        X = _lib.f32(X)
        n, t, d = X.shape
        out = np.empty((n, t, self.hidden), np.float32)
        pad = np.empty((n, t), np.int32)
        _lib.check(self._h, lib.asr_encode(self._h, _lib.fptr(X), n, t, d, _lib.fptr(out), _lib.iptr(pad)))
        return out, pad

    def stack_frames(self, feats, left=3, right=0, sub=3):
        feats = _lib.f32(feats)
        n, dim = feats.shape
        out = np.empty(((n + sub - 1) // sub, dim * (left + right + 1)), np.float32)
        _lib.check(self._h, lib.asr_stack_frames(self._h, _lib.fptr(feats), n, dim, left, right, sub, _lib.fptr(out)))
        return out

    # ---- feature front end (third_party/feat_convert/io/fea.py), one utterance [frames, dim] per call ------
    def _fea(self, feats):
        feats = _lib.f32(feats)
        if feats.ndim != 2 or feats.shape[0] < 1 or feats.shape[1] < 1:
            raise ValueError('features must be a non-empty [frames, dim] matrix')
        return feats

    def fea_add_delt(self, feats):
        """np_fea_add_delt / tf_fea_add_delt (fea.py:61-72,117-128): [n, dim] -> [n, 3*dim]."""
        feats = self._fea(feats)
        n, dim = feats.shape
        out = np.empty((n, 3 * dim), np.float32)
        _lib.check(self._h, lib.asr_fea_add_delt(self._h, _lib.fptr(feats), n, dim, _lib.fptr(out)))
        return out

    def fea_cmvn_global(self, feats, mean, var):
        """tf_fea_cmvn_global (fea.py:130-132): (feats - mean) / var."""
        feats = self._fea(feats)
        n, dim = feats.shape
        mean, var = _lib.f32(mean).reshape(-1), _lib.f32(var).reshape(-1)
        if mean.size != dim or var.size != dim:
            raise ValueError('mean / var must have one value per feature column')
        out = np.empty((n, dim), np.float32)
        _lib.check(self._h, lib.asr_fea_cmvn_global(self._h, _lib.fptr(feats), n, dim, _lib.fptr(mean), _lib.fptr(var),
                                                    _lib.fptr(out)))
        return out

    def fea_cmvn_utt(self, feats):
        """tf_fea_cmvn_utt (fea.py:134-140)."""
        feats = self._fea(feats)
        n, dim = feats.shape
        out = np.empty((n, dim), np.float32)
        _lib.check(self._h, lib.asr_fea_cmvn_utt(self._h, _lib.fptr(feats), n, dim, _lib.fptr(out)))
        return out

    def fea_splice(self, feats, left_num, right_num, sub=1, offset=0, edge=False):
        """tf_fea_splice (fea.py:142-149); sub / offset keep frames offset, offset+sub, ...; edge=True repeats the
        first / last frame as Kaldi's splice-feats does instead of padding with zeros."""
        feats = self._fea(feats)
        n, dim = feats.shape
        if not 0 <= offset < n:
            raise ValueError('offset must name a frame of the utterance')
        out = np.empty(((n - offset + sub - 1) // sub, dim * (left_num + right_num + 1)), np.float32)
        _lib.check(self._h, lib.asr_fea_splice(self._h, _lib.fptr(feats), n, dim, left_num, right_num, sub, offset,
                                               1 if edge else 0, _lib.fptr(out)))
        return out

    # ---- instrumentation ---------------------------------------------------------------------
    def set_profile(self, on):
        """False / 0: off; True / 1: every kernel bracketed and waited for (solo durations); 2: brackets left in the
        queues (durations in the running pipeline, sub-batches side by side)."""
        _lib.check(self._h, lib.asr_set_profile(self._h, int(on)))

    def get_profile(self):
        cap = 64
        names = (C.c_char_p * cap)()
        ms = (C.c_double * cap)()
        launches = (C.c_int64 * cap)()
        flops = (C.c_double * cap)()
        nbytes = (C.c_double * cap)()
        n = C.c_int32(0)
        _lib.check(self._h, lib.asr_get_profile(self._h, cap, names, ms, launches, flops, nbytes, C.byref(n)))
        return [dict(name=names[i].decode(), ms=ms[i], launches=launches[i], flops=flops[i], bytes=nbytes[i])
                for i in range(n.value)]

    def device_bytes(self):
        return int(lib.asr_device_bytes(self._h))

    def range_retries(self):
        """precision f32_f16x3: calls repeated on the bf16x6 kernels because an operand did not fit fp16."""
        return int(lib.asr_range_retries(self._h))


def visible_devices():
    hwqueues.mark_gpu_started()
    return int(lib.asr_device_count())


def default_device():
    """Under `torch.distributed.run` each rank owns the GPU named by LOCAL_RANK."""
    for key in ('ASR_AMD_DEVICE', 'LOCAL_RANK'):
        if os.environ.get(key, '') != '':
            return int(os.environ[key])
    return 0


class EnginePool(object):
    """Several engine handles on ONE GPU - each with its own weight replica, K/V cache, workspaces and streams - driven by
    one host thread each, so that whole batches are decoded side by side (no counterpart in the reference, whose
    `translate` loop runs one `sess.run` at a time, evaluate.py:52-66).  A batch of the reference's size leaves an MI355X
    short of work in two ways: ten-utterance bucketed batches are bound by kernel latency, and a 64-utterance batch has
    launch gaps and tails that other batches' kernels fill.  Measured: 64 x 10 s utterances, beam 10: 291 ms a batch with
    one handle (two half batches on two streams), 239 / 224 ms with two / three handles each running whole-batch kernels, 214 ms
    with four when the process has eight hardware queues (GPU_MAX_HW_QUEUES=8 before the first GPU call; with the default four
    queues a fourth handle shares a queue and loses).  Every batch is decoded exactly as if alone: same hypotheses.

    Unless the YAML sets `test.sub_batches`, the handles of a pool of two or more run each batch as ONE run of
    whole-batch kernels (the batches themselves are what overlaps)."""

    # (class-level defaults: a pool assembled around stand-in handles, as the CPU tests do, has one regime)
    _alone = _together = _regime = (0, 0)
    _regime_users = 0
    _regime_lock = threading.Lock()

    def __init__(self, config, device=0, engines=2):
        n = max(1, int(engines))
        t = config.test
        env_sb = int(os.environ.get('ASR_AMD_SUBBATCH') or 0)
        # the regime of a handle used ALONE (Model.predict, Evaluator.loss, a pool of one): what the YAML says, else the
        # library's defaults (K-slices and tiles that fill the chip by themselves, two half batches from 320 beam rows)
        self._alone = (1 if t.shared_gpu else 0, int(t.sub_batches or env_sb or 0))
        # ... and while the pool's handles decode side by side (map): fewer K-slices, every batch ONE run of kernels
        self._together = (1 if (t.shared_gpu is None or t.shared_gpu) else 0, int(t.sub_batches or env_sb or 1))
        self._regime = self._alone
        self._regime_lock = threading.Lock()
        self._regime_users = 0
        if n >= 4:
            # a hardware queue per handle (asr_amd/hwqueues.py); has to happen before the process first touches the GPU
            self.hw_queues = hwqueues.ensure()
        else:
            self.hw_queues = {'value': hwqueues.current(), 'source': 'not changed (fewer than four handles)', 'late': False}
        self.engines = [Engine(config, device) for _ in range(n)]

    def load_weights(self, variables):
        """The first handle packs and uploads the checkpoint; the others share its device copy (read-only planes)."""
        self.engines[0].load_weights(variables)
        for e in self.engines[1:]:
            e.share_weights(self.engines[0])

    def device_bytes(self):
        return sum(e.device_bytes() for e in self.engines)

    def close(self):
        for e in self.engines:
            e.close()

    def together(self):
        """Context manager: the pool's handles are in the shared-GPU regime inside (nestable, thread-safe); `map` uses it.
        Outside, every handle is in the regime of a handle that has the GPU to itself."""
        pool = self

        class _Together(object):
            def __enter__(self):
                with pool._regime_lock:
                    pool._regime_users += 1
                    if pool._regime_users == 1 and len(pool.engines) > 1 and pool._together != pool._alone:
                        for e in pool.engines:
                            e.set_sharing(*pool._together)
                        pool._regime = pool._together
                return pool

            def __exit__(self, *exc):
                with pool._regime_lock:
                    pool._regime_users -= 1
                    if pool._regime_users == 0 and pool._regime != pool._alone:
                        for e in pool.engines:
                            if getattr(e, '_h', None):
                                e.set_sharing(*pool._alone)
                        pool._regime = pool._alone
                return False
        return _Together()

    def map(self, call, items):
        """Ordered iterator over call(engine, item) for item in items, with as many items in flight as there are
        handles (items are drawn lazily, at most that many ahead of the results consumed).  Every handle is driven by
        a host thread of its own for the whole of the iteration."""
        if len(self.engines) == 1:
            for it in items:
                yield call(self.engines[0], it)
            return
        n = len(self.engines)
        source = iter(items)
        cond = threading.Condition()
        st = {'next_in': 0, 'next_out': 0, 'exhausted': False, 'stop': False, 'results': {}}

        class _Failed(object):
            def __init__(self, ex):
                self.ex = ex

        def worker(e):
            while True:
                with cond:
                    while not st['stop'] and not st['exhausted'] and st['next_in'] - st['next_out'] >= n:
                        cond.wait()
                    if st['stop'] or st['exhausted']:
                        return
                    try:
                        item = next(source)
                    except StopIteration:
                        st['exhausted'] = True
                        cond.notify_all()
                        return
                    except Exception as ex:          # the caller's generator failed: surfaces at this place in the order
                        st['results'][st['next_in']] = _Failed(ex)
                        st['next_in'] += 1
                        st['exhausted'] = True
                        cond.notify_all()
                        return
                    idx = st['next_in']
                    st['next_in'] += 1
                try:
                    r = call(e, item)
                except Exception as ex:
                    r = _Failed(ex)
                with cond:
                    st['results'][idx] = r
                    cond.notify_all()
        threads = [threading.Thread(target=worker, args=(e,), daemon=True) for e in self.engines]
        regime = self.together()
        regime.__enter__()
        for th in threads:
            th.start()
        try:
            while True:
                with cond:
                    while st['next_out'] not in st['results'] and not (st['exhausted'] and st['next_out'] >= st['next_in']):
                        cond.wait()
                    if st['next_out'] not in st['results']:
                        return
                    r = st['results'].pop(st['next_out'])
                    st['next_out'] += 1
                    cond.notify_all()
                if isinstance(r, _Failed):
                    raise r.ex
                yield r
        finally:
            with cond:
                st['stop'] = True
                cond.notify_all()
            for th in threads:
                th.join()
            regime.__exit__(None, None, None)

    def decode_many(self, batches):
        """Y of every X in `batches`, in order."""
        return self.map(lambda e, X: e.decode(X), batches)


class Model(object):
    """Inference-only counterpart of model.Model (model.py:18-45,162-204).

    `num_gpus` is the number of towers: the batch is split on dim 0 exactly as
    `expand_feed_dict` does (utils.py:420-443), each shard is decoded on its own (its own padded
    length T, Q7/Q10), predictions are padded with 3 to the longest and concatenated in tower
    order.  Towers are placed round-robin on the GPUs visible to this process (`devices`);
    towers on different GPUs run concurrently."""

    def __init__(self, config, num_gpus, devices=None):
        self._config = config
        self._num_towers = max(int(num_gpus or 0), 1)            # model.py:27: 0 GPUs -> one CPU tower
        # build extension (not a reference key): engine handles per GPU, see EnginePool.  Unset, with one tower: the measured
        # optimum for whole test sets (`predict_batches`, i.e. Evaluator.translate) - four handles when the process has, or
        # can still get, eight hardware queues, three on the runtime's default four; `predict` / `loss` use the first
        # handle alone in either case.  Decided BEFORE the first GPU call of this process (asr_amd/hwqueues.py).
        epg = (config.test.engines_per_gpu if config.test else None) or os.environ.get('ASR_AMD_ENGINES')
        if epg:
            self._engines_per_gpu = max(1, int(epg))
        else:
            self._engines_per_gpu = None                             # decided below, once the devices are known
        if devices is None:
            if 'LOCAL_RANK' in os.environ or 'ASR_AMD_DEVICE' in os.environ:
                devices = [default_device()]
            else:
                devices = list(range(max(1, min(self._num_towers, visible_devices()))))
        self._devices = list(devices)
        # build extension `test.joint_towers` (default true; env ASR_AMD_JOINT_TOWERS=0): towers that share a GPU are decoded by one
        # joint call; false = one call per tower, one after the other
        jt = config.test.joint_towers if config.test else None
        self._joint_towers = (os.environ.get('ASR_AMD_JOINT_TOWERS', '1') != '0') if jt is None else bool(jt)
        if self._engines_per_gpu is None:
            if len(self._devices) == 1:                              # one GPU for this process, however many towers the YAML names
                q = hwqueues.ensure()
                self._engines_per_gpu = 4 if q['value'] >= hwqueues.WANTED else 3
                logging.info('asr_amd: test.engines_per_gpu unset: %d engine handles per GPU on %d hardware queues (%s)',
                             self._engines_per_gpu, q['value'], q['source'])
            else:
                self._engines_per_gpu = 1
        self.pools = []
        self.engines = []
        self.encoder_scope = 'encoder'
        self.decoder_scope = 'decoder'

    def build_test_model(self, reuse=None):
        logging.info('Build test model.')
        if not self.engines:
            self.pools = [EnginePool(self._config, d, self._engines_per_gpu) for d in self._devices]
            self.engines = [p.engines[0] for p in self.pools]          # predict / loss: the first handle of every GPU

    def restore(self, variables):
        for p in self.pools:
            p.load_weights(variables)

    def close(self):
        for p in self.pools:
            p.close()
        self.pools, self.engines = [], []

    def predict_batches(self, batches):
        """predict(X) for every X of `batches`, in order.  One tower on one GPU with `test.engines_per_gpu` >= 2: that many
        batches are in flight at a time (EnginePool); otherwise one after the other, like evaluate.py:52-66."""
        merge = int((self._config.test.merge_utts if self._config.test else None) or os.environ.get('ASR_AMD_MERGE_UTTS') or 0)
        if merge > 0 and self._num_towers == 1:
            # build extension (not a reference key) `test.merge_utts`: consecutive batches of the bucketed reader are joined
            # into joint decodes of up to that many utterances (Engine.decode_groups: every batch keeps its own padded length
            # wherever it enters the arithmetic, the decode steps run over all of them), results handed back batch by batch
            def jobs():
                cur, n = [], 0
                for X in batches:
                    X = np.asarray(X, dtype=np.float32)
                    if cur and (n + X.shape[0] > merge or X.shape[0] == 0):
                        yield cur
                        cur, n = [], 0
                    cur.append(X)
                    n += X.shape[0]
                    if X.shape[0] == 0:
                        yield cur
                        cur, n = [], 0
                if cur:
                    yield cur

            def call(e, Xs):
                return [e.decode(Xs[0])] if len(Xs) == 1 else e.decode_groups(Xs)
            if len(self.pools) == 1 and len(self.pools[0].engines) > 1:
                done = self.pools[0].map(call, jobs())
            else:
                done = (call(self.engines[0], Xs) for Xs in jobs())
            return (Y for Ys in done for Y in Ys)
        if self._num_towers == 1 and len(self.pools) == 1 and len(self.pools[0].engines) > 1:
            pair = self._config.test.pair_batches if self._config.test else None
            if pair is None:
                pair = os.environ.get('ASR_AMD_PAIR_BATCHES', '1') != '0'
            if not pair:
                return self.pools[0].map(lambda e, X: e.decode(np.asarray(X, dtype=np.float32)), batches)
            # Whole batches (PAIR_MIN_ROWS beam rows or more each) are decoded two per library call (Engine.decode_groups): every weight
            # plane is streamed once for both.  From that row count on the handles of a pool choose the same K-slices whatever the
            # number of rows, so every float32 sum is taken in the same order as in a call of the batch alone: the SAME hypotheses,
            # bit for bit (`test.pair_batches: false` / ASR_AMD_PAIR_BATCHES=0 turns it off).  Smaller batches go one per call.
            beam = int(self._config.test.beam_size)

            def paired():
                held = None
                for X in batches:
                    X = np.asarray(X, dtype=np.float32)
                    whole = X.shape[0] * beam >= PAIR_MIN_ROWS
                    if held is not None:
                        if whole:
                            yield [held, X]
                            held = None
                            continue
                        yield [held]
                        held = None
                    if whole:
                        held = X
                    else:
                        yield [X]
                if held is not None:
                    yield [held]

            def call(e, Xs):
                return [e.decode(Xs[0])] if len(Xs) == 1 else e.decode_groups(Xs)
            return (Y for Ys in self.pools[0].map(call, paired()) for Y in Ys)
        if self._num_towers > 1 and len(self.pools) == 1 and len(self.pools[0].engines) > 1:
            # the YAML names several towers (the reference's configs ship `test.num_gpus: 4`) and this process has one GPU: every
            # batch is still split into its towers' shards (below), and that many batches are in flight over the pool's handles
            return self.pools[0].map(lambda e, X: self._predict_on([e], X), batches)
        return (self.predict(X) for X in batches)

    def predict(self, X):
        """== sess.run(self.prediction, expand_feed_dict({self.src_pls: X}))."""
        return self._predict_on(self.engines, X)

    def _predict_on(self, engines, X):
        """model.py:168-201 on `engines` (one per GPU): dim 0 split into the towers' shards exactly as expand_feed_dict does,
        every shard decoded as a beam search of its own (its own loop end and prediction length), predictions padded with 3 to
        the longest and concatenated in tower order.  Towers that land on the same GPU are decoded by ONE joint call
        (Engine.decode_groups: the shards stay groups of their own wherever that enters the arithmetic, the row-wise
        kernels run over all of them - one pass over the weights instead of one per tower)."""
        X = np.asarray(X, dtype=np.float32)
        bounds = shard_bounds(X.shape[0], self._num_towers)
        preds = [None] * self._num_towers
        by_engine = {}
        for i, (a, b) in enumerate(bounds):
            by_engine.setdefault(i % len(engines), []).append((i, a, b))

        def work(ei, items):
            full = [(i, a, b) for (i, a, b) in items if b > a]
            for (i, a, b) in items:
                if b <= a:
                    preds[i] = np.zeros((0, 0), np.int32)                                            # model.py:184-187
            if len(full) == 1 or not self._joint_towers:
                for (i, a, b) in full:
                    preds[i] = engines[ei].decode(X[a:b])
            elif full:
                for (i, a, b), Y in zip(full, engines[ei].decode_groups([X[a:b] for (_, a, b) in full])):
                    preds[i] = Y

        if len(by_engine) == 1:
            work(*next(iter(by_engine.items())))
        else:
            errs = []

            def guarded(ei, items):
                try:
                    work(ei, items)
                except Exception as ex:      # re-raised below
                    errs.append(ex)
            threads = [threading.Thread(target=guarded, args=kv) for kv in by_engine.items()]
            for th in threads:
                th.start()
            for th in threads:
                th.join()
            if errs:
                raise errs[0]
        max_length = max(p.shape[1] for p in preds)                                                  # model.py:192
        preds = [np.concatenate([p, np.full((p.shape[0], max_length - p.shape[1]), 3, np.int32)], axis=1)
                 for p in preds]                                                                     # :194-200
        return np.concatenate(preds, axis=0)                                                         # :201


    def loss(self, X, Y):
        """== sess.run(self.loss_sum, expand_feed_dict({self.src_pls: X, self.dst_pls: Y})) (model.py:168-190)."""
        X = np.asarray(X, dtype=np.float32)
        Y = np.asarray(Y, dtype=np.int32)
        total = 0.0
        for i, (a, b) in enumerate(shard_bounds(X.shape[0], self._num_towers)):
            if b > a:
                total += self.engines[i % len(self.engines)].loss(X[a:b], Y[a:b])
        return total


class Transformer(Model):
    """config.model == 'Transformer' (model.py:434-442)."""
    pass
