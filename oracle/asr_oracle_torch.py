"""SECOND, independent CPU restatement of the reference's batched inference path, in torch-CPU.

TEST INFRASTRUCTURE ONLY (tests/ may import it; the product path and bench.py's timed region never do).
PARITY UNPINNED: the reference is Python-2 / TensorFlow-1.x graph code that cannot run in the build image and
ships no fixtures, so neither this file nor oracle/asr_oracle.py can be checked against reference outputs.  What
this file adds is protection against a shared misreading: it was written from the reference's source lines
(cited below, relative to /root/reference/transformer/) and SURVEY.md appendix A, not from the numpy restatement,
uses a different array library, and follows the reference's own data flow literally (tile the encoder output over
beams, block-output cache [R, i, L, H], K/V re-projected every step, tf.gather of the cache by parent).
tests/test_oracle_cross.py requires the two restatements to agree.

Conventions: W = {TF variable name: numpy array}, cfg = any object with the YAML key names as
attributes (hidden_units, num_heads, num_blocks, dst_vocab_size, ff_activation, scale_embedding,
tie_embedding_and_softmax, beam_size, max_target_length, lp_alpha).
"""
import math

import numpy as np
import torch

NEG = -1e9     # common_attention.py:189,202
INF = 1e10     # model.py:224


def _t(a, dtype):
    return torch.as_tensor(np.asarray(a), dtype=dtype)


class _Vars(object):
    """Variable store with tf.variable_scope-like prefixes."""

    def __init__(self, W, dtype):
        self.W, self.dtype = W, dtype

    def get(self, name):
        return _t(self.W[name], self.dtype)


# ---- utils.py:534-561 -------------------------------------------------------------------------
def dense(v, scope, x, use_bias=True):
    """reshape(x, [-1, in]) . kernel[out, in]^T (+ bias)  (utils.py:546-557)."""
    w = v.get(scope + '/kernel')
    shape = x.shape
    y = x.reshape(-1, shape[-1]) @ w.t()
    if use_bias:
        y = y + v.get(scope + '/bias')
    return y.reshape(*shape[:-1], w.shape[0])


_ACT1 = {'relu': torch.relu, 'sigmoid': torch.sigmoid, 'tanh': torch.tanh, 'swish': lambda x: x * torch.sigmoid(x)}


def ff_hidden(v, scope, x, activation):
    """utils.py:564-568; a 2-argument activation (glu) makes two denses arg1 / arg2 (utils.py:559-561; model.py:441)."""
    if activation == 'glu':
        a1 = dense(v, scope + '/ff_hidden/dense/arg1', x)
        a2 = dense(v, scope + '/ff_hidden/dense/arg2', x)
        h = a1 * torch.sigmoid(a2)
    else:
        h = _ACT1[activation](dense(v, scope + '/ff_hidden/dense', x))
    return dense(v, scope + '/ff_hidden/dense_1', h)


# ---- common_layers.py:333-371 -----------------------------------------------------------------
def layer_norm(v, scope, x, eps=1e-6):
    mean = x.mean(dim=-1, keepdim=True)
    var = ((x - mean) ** 2).mean(dim=-1, keepdim=True)
    norm = (x - mean) * torch.rsqrt(var + eps)
    return norm * v.get(scope + '/layer_norm_scale') + v.get(scope + '/layer_norm_bias')


def residual(v, scope, inputs, outputs):
    """utils.py:494-507 with dropout = identity."""
    return layer_norm(v, scope, inputs + outputs)


# ---- tf.contrib.layers.layer_norm, TF1 defaults (model.py:453) ------------------
def contrib_layer_norm(v, scope, x):
    """moments over every axis but the batch (T x H), variance_epsilon 1e-12, gamma / beta [H];
    tf.nn.batch_normalization: x * inv + (beta - mean * inv), inv = rsqrt(var + eps) * gamma."""
    mean = x.mean(dim=(1, 2), keepdim=True)
    var = ((x - mean) ** 2).mean(dim=(1, 2), keepdim=True)
    inv = torch.rsqrt(var + 1e-12) * v.get(scope + '/gamma')
    return x * inv + (v.get(scope + '/beta') - mean * inv)


# ---- common_attention.py ------------------------------------------------------------------------
def timing_signal(length, channels, dtype):
    """common_attention.py:56-68, max_timescale 1e5 (Q2).  float32: every op rounds to float32 as the graph does."""
    nts = channels // 2
    position = torch.arange(length, dtype=dtype)
    # Python float / float32 tensor in TF: the float is converted to a float32 tensor first, then divided in float32
    inc = torch.tensor(math.log(1.0e5 / 1.0), dtype=dtype) / (torch.tensor(float(nts), dtype=dtype) - 1)
    # TF's float32 exp / sin / cos kernels cannot be reproduced here: each elementary function is taken correctly
    # rounded (evaluated in float64, rounded once), the multiplications and the division round in `dtype` like the graph
    inv = 1.0 * torch.exp((torch.arange(nts, dtype=dtype) * -inc).double()).to(dtype)
    scaled = position[:, None] * inv[None, :]
    sig = torch.cat([torch.sin(scaled.double()), torch.cos(scaled.double())], dim=1).to(dtype)
    if channels % 2:
        sig = torch.nn.functional.pad(sig, (0, 1))
    return sig.reshape(1, length, channels)


def split_heads(x, n):          # :243-253   [B, len, H] -> [B, n, len, H/n]
    b, ln, h = x.shape
    return x.reshape(b, ln, n, h // n).permute(0, 2, 1, 3)


def combine_heads(x):           # :256-265
    b, n, ln, d = x.shape
    return x.permute(0, 2, 1, 3).reshape(b, ln, n * d)


def dot_product_attention(q, k, v, bias):   # :295-334
    logits = q @ k.transpose(-1, -2)
    if bias is not None:
        logits = logits + bias
    return torch.softmax(logits, dim=-1) @ v


def multihead_attention(v, scope, query, memory, bias, heads, reserve_last=False):
    """utils.py:571-633."""
    H = query.shape[-1]
    if memory is None:
        combined = dense(v, scope + '/qkv_transform', query)
        q, k, val = combined[..., :H], combined[..., H:2 * H], combined[..., 2 * H:]
    else:
        q = dense(v, scope + '/q_transform', query)
        combined = dense(v, scope + '/kv_transform', memory)
        k, val = combined[..., :H], combined[..., H:]
    if reserve_last:
        q = q[:, -1:, :]
    q, k, val = split_heads(q, heads), split_heads(k, heads), split_heads(val, heads)
    q = q * (H // heads) ** -0.5                       # after the bias (Q3)
    x = combine_heads(dot_product_attention(q, k, val, bias))
    return dense(v, scope + '/output_transform', x)


def padding_bias(pad, dtype):   # attention_bias_ignore_padding :193-203  -> [B, 1, 1, Tk]
    return (pad.to(dtype) * NEG)[:, None, None, :]


# ---- model.py:444-490 -----------------------------------------------------------------------------
def encoder(X, W, cfg, dtype=torch.float32):
    v = _Vars(W, dtype)
    x_in = _t(X, dtype)
    pad = x_in.abs().sum(dim=-1) == 0.0                                      # :450
    x = dense(v, 'encoder/src_change', x_in)                                 # :451
    x = contrib_layer_norm(v, 'encoder/LayerNorm', x)                        # :453
    x = x + timing_signal(x.shape[1], x.shape[2], dtype)                     # :456
    bias = padding_bias(pad, dtype)
    for i in range(cfg.num_blocks):
        p = 'encoder/block_%d' % i
        a = multihead_attention(v, p + '/encoder_self_attention', x, None, bias, cfg.num_heads)
        x = residual(v, p + '/layer_norm', x, a)                             # :466-478
        x = residual(v, p + '/layer_norm_1', x, ff_hidden(v, p, x, cfg.ff_activation))   # :481-487
    x = x * (1.0 - pad.to(dtype))[..., None]                                 # :489
    return x, pad


# ---- model.py:559-633 (literal: block-output cache, everything re-projected each step) ---------------
def decoder_with_caching(v, cfg, preds, cache, enc, dtype):
    H = cfg.hidden_units
    enc_pad = enc.abs().sum(dim=-1) == 0.0                                   # :566
    enc_bias = padding_bias(enc_pad, dtype)
    emb = v.get('decoder/dst_embedding/kernel')
    y = emb[preds.long()]                                                    # utils.py:528
    if cfg.scale_embedding:
        y = y * (H ** 0.5)
    y = y + timing_signal(y.shape[1], H, dtype)                              # :576
    new_cache = []
    for i in range(cfg.num_blocks):
        p = 'decoder/block_%d' % i
        s = multihead_attention(v, p + '/decoder_self_attention', y, None, None, cfg.num_heads, reserve_last=True)
        z = residual(v, p + '/layer_norm', y[:, -1:, :], s)                  # :588-601
        c = multihead_attention(v, p + '/decoder_vanilla_attention', z, enc, enc_bias, cfg.num_heads, reserve_last=True)
        z = residual(v, p + '/layer_norm_1', z, c)                           # :604-617
        z = residual(v, p + '/layer_norm_2', z, ff_hidden(v, p, z, cfg.ff_activation))   # :620-626
        y = torch.cat([cache[:, :, i, :], z], dim=1)                         # :628
        new_cache.append(y[:, :, None, :])
    return y, torch.cat(new_cache, dim=2)


def test_output(v, cfg, dec_out):
    """model.py:338-348.  top_k(sorted=False): candidate order is implementation-defined in TF (Q5); emitted
    here by (value desc, index asc)."""
    name = 'decoder/dst_embedding' if cfg.tie_embedding_and_softmax else 'decoder/softmax'
    logits = dense(v, name, dec_out[:, -1], use_bias=False)
    z = torch.log_softmax(logits, dim=-1)
    K = cfg.beam_size
    order = torch.argsort(-z, dim=-1, stable=True)[:, :K]
    return logits, order.to(torch.int32), torch.gather(z, 1, order)


def beam_search(enc, W, cfg, dtype=torch.float32, trace=None):
    """model.py:221-336."""
    v = _Vars(W, dtype)
    enc = _t(enc, dtype)
    K, B = cfg.beam_size, enc.shape[0]
    R = B * K
    enc = enc[:, None].repeat(1, K, 1, 1).reshape(R, enc.shape[1], enc.shape[2])          # :257-261
    preds = torch.full((R, 1), 2, dtype=torch.int32)                                      # :263
    scores = torch.tensor([0.0] + [-INF] * (K - 1), dtype=torch.float32).repeat(B)        # :264-265 (float32 in the graph)
    bias = torch.zeros(R, dtype=torch.bool)
    cache = torch.zeros(R, 0, cfg.num_blocks, cfg.hidden_units, dtype=dtype)
    limit = min(enc.shape[1] + 50, cfg.max_target_length)                                 # :310
    i = 0
    sdtype = torch.float32 if dtype == torch.float32 else dtype
    scores = scores.to(sdtype)
    while bool((~bias).any()) and i <= limit:                                             # :305-312
        i += 1
        dec_out, cache = decoder_with_caching(v, cfg, preds, cache, enc, dtype)
        logits, k_preds, k_scores = test_output(v, cfg, dec_out)
        bi = bias.to(torch.int32)
        k_preds = k_preds * (1 - bi[:, None]) + bi[:, None] * 3                           # :252-253
        bf = bias.to(sdtype)
        b = torch.tensor([0.0] + [-INF] * (K - 1), dtype=sdtype)[None, :].repeat(R, 1)
        k_scores = k_scores.to(sdtype) * (1 - bf[:, None]) + b * bf[:, None]              # :237-240
        tot = (scores[:, None] + k_scores).reshape(B, K * K)                              # :284-285
        order = torch.argsort(-tot, dim=-1, stable=True)[:, :K]                           # sorted top_k, ties -> lower index
        scores = torch.gather(tot, 1, order).reshape(-1)
        k_idx = (torch.arange(B)[:, None] * (K * K) + order).reshape(-1)                  # :290-292
        last = k_preds.reshape(-1)[k_idx]
        parent = torch.div(k_idx, K, rounding_mode='floor')                               # Py2 int / (Q12)
        preds = torch.cat([preds[parent], last[:, None]], dim=1)                          # :296,298
        cache = cache[parent]                                                             # :297
        bias = preds[:, -1] == 3                                                          # :301
        if trace is not None:
            trace.setdefault('logits', []).append(logits.numpy().copy())
            trace.setdefault('tokens', []).append(last.numpy().copy())
            trace.setdefault('parents', []).append(parent.numpy().copy())
            trace.setdefault('scores', []).append(scores.numpy().copy())
    scores = scores.reshape(B, K)
    preds3 = preds.reshape(B, K, -1)
    lengths = (preds3 != 3).to(sdtype).sum(dim=-1)                                        # :327 (counts the leading <S>)
    lp = ((5 + lengths) / (5 + 1)) ** cfg.lp_alpha                                        # :328
    best = torch.argmax(scores / lp, dim=-1) + torch.arange(B) * K                        # :329-331
    return preds[best][:, 1:].numpy().astype(np.int32)                                    # :334-335


def build_test_model(X, W, cfg, num_gpus=1, dtype=torch.float32):
    """model.py:162-204 with the feeder's dim-0 split (utils.py:420-443): per-tower encoder + beam search,
    empty towers give [0, 0], predictions padded with 3 to the longest and concatenated."""
    X = np.asarray(X)
    n = X.shape[0]
    span, rem = n // num_gpus, n % num_gpus
    outs, base = [], 0
    for g in range(num_gpus):
        end = base + span + (1 if g < rem else 0)
        if end > base:
            enc, _ = encoder(X[base:end], W, cfg, dtype)
            outs.append(beam_search(enc, W, cfg, dtype))
        else:
            outs.append(np.zeros((0, 0), np.int32))
        base = end
    mx = max(o.shape[1] for o in outs)
    outs = [np.concatenate([o, np.full((o.shape[0], mx - o.shape[1]), 3, np.int32)], axis=1) for o in outs]
    return np.concatenate(outs, axis=0)
