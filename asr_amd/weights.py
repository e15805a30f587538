"""Checkpoint side of the drop-in: variable names/shapes of the inference graph, `latest_checkpoint`,
checkpoint loading (npz with TF variable names, or a TF-V2 bundle via tf_bundle.py) and the seeded
random initialiser used for synthetic benchmarks.

Reference: `model.saver.restore(sess, tf.train.latest_checkpoint(config.model_dir))`
(evaluate.py:36), variable creation sites of SURVEY.md A.1 (utils.py:527,551,554;
common_layers.py:362-365; model.py:453).
"""
from __future__ import annotations

import math
import os
import re

import numpy as np


def variable_shapes(config):
    """TF variable name -> shape for the test graph (`Transformer`)."""
    H, D, V, L = int(config.hidden_units), int(config.train.input_dim), int(config.dst_vocab_size), int(config.num_blocks)
    F = 4 * H
    glu = (config.ff_activation == 'glu')
    s = {'encoder/src_change/kernel': (H, D), 'encoder/src_change/bias': (H,),
         'encoder/LayerNorm/beta': (H,), 'encoder/LayerNorm/gamma': (H,)}

    def ffn(p):
        if glu:
            for a in ('arg1', 'arg2'):
                s['%s/ff_hidden/dense/%s/kernel' % (p, a)] = (F, H)
                s['%s/ff_hidden/dense/%s/bias' % (p, a)] = (F,)
        else:
            s[p + '/ff_hidden/dense/kernel'] = (F, H)
            s[p + '/ff_hidden/dense/bias'] = (F,)
        s[p + '/ff_hidden/dense_1/kernel'] = (H, F)
        s[p + '/ff_hidden/dense_1/bias'] = (H,)

    def attn(p, name, kind):
        if kind == 'self':
            s['%s/%s/qkv_transform/kernel' % (p, name)] = (3 * H, H)
            s['%s/%s/qkv_transform/bias' % (p, name)] = (3 * H,)
        else:
            s['%s/%s/q_transform/kernel' % (p, name)] = (H, H)
            s['%s/%s/q_transform/bias' % (p, name)] = (H,)
            s['%s/%s/kv_transform/kernel' % (p, name)] = (2 * H, H)
            s['%s/%s/kv_transform/bias' % (p, name)] = (2 * H,)
        s['%s/%s/output_transform/kernel' % (p, name)] = (H, H)
        s['%s/%s/output_transform/bias' % (p, name)] = (H,)

    def lns(p, n):
        for i in range(n):
            ln = 'layer_norm' if i == 0 else 'layer_norm_%d' % i
            s['%s/%s/layer_norm_scale' % (p, ln)] = (H,)
            s['%s/%s/layer_norm_bias' % (p, ln)] = (H,)

    for i in range(L):
        p = 'encoder/block_%d' % i
        attn(p, 'encoder_self_attention', 'self')
        ffn(p)
        lns(p, 2)
    s['decoder/dst_embedding/kernel'] = (V, H)
    if not config.tie_embedding_and_softmax:
        s['decoder/softmax/kernel'] = (V, H)
    for i in range(L):
        p = 'decoder/block_%d' % i
        attn(p, 'decoder_self_attention', 'self')
        attn(p, 'decoder_vanilla_attention', 'cross')
        ffn(p)
        lns(p, 3)
    return s


def random_init(config, seed=0):
    """What an un-restored TF1 graph would hold: glorot-uniform kernels
    (limit sqrt(6/(fan_in+fan_out)) over the two kernel dims), zero biases, unit LN scales -- drawn
    from this build's own seeded generator (TF's RNG stream is not reproducible)."""
    rng = np.random.default_rng(seed)
    W = {}
    for name, shape in sorted(variable_shapes(config).items()):
        if name.endswith('/kernel'):
            limit = math.sqrt(6.0 / (shape[0] + shape[1]))
            W[name] = rng.uniform(-limit, limit, size=shape).astype(np.float32)
        elif name.endswith('gamma') or name.endswith('layer_norm_scale'):
            W[name] = np.ones(shape, np.float32)
        else:
            W[name] = np.zeros(shape, np.float32)
    return W


def latest_checkpoint(model_dir):
    """tf.train.latest_checkpoint: the path named by `model_checkpoint_path` in `<dir>/checkpoint`
    (after avg_checkpoints.py that is `averaged.ckpt-0`).  Falls back to a lone `*.npz`."""
    state = os.path.join(model_dir, 'checkpoint')
    if os.path.exists(state):
        with open(state) as f:
            for line in f:
                m = re.match(r'\s*model_checkpoint_path:\s*"(.*)"\s*$', line)
                if m:
                    p = m.group(1)
                    return p if os.path.isabs(p) else os.path.join(model_dir, p)
    if os.path.isdir(model_dir):
        npz = sorted(f for f in os.listdir(model_dir) if f.endswith('.npz'))
        if npz:
            return os.path.join(model_dir, npz[-1])
    return None


_SKIP = re.compile(r'(^global_step)|(/Adam(_1)?$)|(^beta[12]_power$)')


def load_checkpoint(path):
    """Checkpoint prefix or file -> {TF variable name: float32 array}, optimizer slots dropped.
    Accepts `<prefix>.npz` / `*.npz` (numpy, allow_pickle=False) and TF-V2 bundles
    (`<prefix>.index` + `<prefix>.data-*`)."""
    if path is None:
        raise IOError('no checkpoint found')
    if path.endswith('.npz') or os.path.exists(path + '.npz'):
        f = path if path.endswith('.npz') else path + '.npz'
        with np.load(f, allow_pickle=False) as z:
            return {k: np.asarray(z[k], np.float32) for k in z.files if not _SKIP.search(k)}
    if os.path.exists(path + '.index'):
        from . import tf_bundle
        out = {}
        for name, arr in tf_bundle.read_bundle(path).items():
            if _SKIP.search(name) or arr.dtype.kind != 'f':
                continue
            out[name] = np.asarray(arr, np.float32)
        return out
    raise IOError('unsupported checkpoint: %s' % path)


def save_npz(path, variables):
    np.savez(path if path.endswith('.npz') else path + '.npz', **variables)
