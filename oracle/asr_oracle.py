"""CPU ORACLE (test infrastructure only -- NOT part of the product path).

A numpy restatement of the batched inference path of shiyuzh2007/ASR
(`transformer/evaluate.py` -> `Model.build_test_model` -> encoder -> in-graph beam search).
Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import
this file; `asr_amd/` never does.

PARITY UNPINNED: the reference is Python-2 / TensorFlow-1.x graph code with no tests, no
golden vectors and no fixtures, and neither Python 2 nor TensorFlow
exists in the build container, so the reference cannot be executed to pin this file.  What
pins it instead: the analytic known-answer tests of SURVEY.md 8(c) (tests/test_oracle_kat.py),
agreement of the float32 and float64 evaluations, and token identity between the two decode
algorithms below (the reference's block-output cache and the K/V-cache formulation the GPU
uses).  The TensorFlow-side numerics (contrib layer_norm, top_k order, softmax) are restated
from the TF 1.x sources as documented in SURVEY.md A.3 (Q1..Q13).

All file:line citations are relative to /root/reference/transformer/.
"""
from __future__ import annotations

import math

import numpy as np

PAD_INDEX, UNK_INDEX, BOS_INDEX, EOS_INDEX = 0, 1, 2, 3   # utils.py:16-19
NEG_INF_BEAM = 1e10                                        # model.py:224
MASK_BIAS = -1e9                                           # common_attention.py:189,202


# --------------------------------------------------------------------------------------
# front end
# --------------------------------------------------------------------------------------
def tf_fea_splice(features, left_num, right_num):
    """third_party/feat_convert/io/fea.py:142-149 -- zero-pad `left_num` rows before and
    `right_num` after, out[t] = concat(x[t-left] .. x[t+right]) along the feature axis."""
    features = np.asarray(features)
    t, d = features.shape
    pp = np.concatenate([np.zeros((left_num, d), features.dtype), features,
                         np.zeros((right_num, d), features.dtype)], axis=0)
    return np.concatenate([pp[i:i + t] for i in range(left_num + right_num + 1)], axis=1)


def frame_stack(features, left=3, right=0, sub=3):
    """'left3_sub3' front end: splice, then keep frames 0, sub, 2*sub, ... (SURVEY 8a1;
    the subsampling is implied by the scp file names, config...char...yaml:24,57)."""
    return tf_fea_splice(features, left, right)[::sub]


def splice_general(features, left_num, right_num, sub=1, offset=0, edge=False):
    """tf_fea_splice generalised the way asr_fea_splice is: frames offset, offset+sub, ... are kept; with `edge`
    the context frames outside the utterance repeat the first / last frame (Kaldi splice-feats) instead of
    being zeros (fea.py:142-149)."""
    x = np.asarray(features)
    t, d = x.shape
    cols = []
    for c in range(-left_num, right_num + 1):
        idx = np.arange(t) + c
        inside = (idx >= 0) & (idx < t)
        block = x[np.clip(idx, 0, t - 1)]
        cols.append(block if edge else np.where(inside[:, None], block, np.zeros((), x.dtype)))
    return np.concatenate(cols, axis=1)[offset::sub]


def _shifted(x, o):
    """x[t+o] with the frame index clamped to the utterance: what fea.py's repeated one-frame "symmetric" pads
    (fea.py:21-22,34-37) amount to, since mirroring an already padded matrix copies the edge frame again."""
    t = x.shape[0]
    return x[np.clip(np.arange(t) + o, 0, t - 1)]


def fea_delt1(features):
    """np_kaldi_fea_delt1 / tf_kaldi_fea_delt1 (fea.py:20-31,75-86), evaluated in the dtype of `features`."""
    x = np.asarray(features)
    k1, k2 = x.dtype.type(0.1), x.dtype.type(0.2)
    return (_shifted(x, 1) - _shifted(x, -1)) * k1 + (_shifted(x, 2) - _shifted(x, -2)) * k2


def fea_delt2(features):
    """np_kaldi_fea_delt2 / tf_kaldi_fea_delt2 (fea.py:33-52,88-107), same operation order."""
    x = np.asarray(features)
    f = x.dtype.type
    s = lambda o: _shifted(x, o)
    out = f(-0.1) * s(0) - f(0.04) * (s(-1) + s(1))
    out = out + f(0.01) * (s(-2) + s(2))
    out = out + f(0.04) * (((s(-3) + s(-4)) + s(4)) + s(3))
    return out


def fea_add_delt(features):
    """np_fea_add_delt / tf_fea_add_delt (fea.py:61-72,117-128): static | delta | delta-delta."""
    x = np.asarray(features)
    return np.concatenate([x, fea_delt1(x), fea_delt2(x)], axis=1)


def fea_cmvn_global(features, mean, var):
    """tf_fea_cmvn_global (fea.py:130-132); note the division is by `var` itself."""
    x = np.asarray(features)
    return (x - np.asarray(mean, x.dtype)) / np.asarray(var, x.dtype)


def fea_cmvn_utt(features):
    """tf_fea_cmvn_utt (fea.py:134-140)."""
    x = np.asarray(features)
    mean = x.mean(axis=0)
    var = np.square(x).mean(axis=0) - mean * mean
    return (x - mean) * (x.dtype.type(1) / np.sqrt(var + x.dtype.type(1e-12)))


# --------------------------------------------------------------------------------------
# layer library
# --------------------------------------------------------------------------------------
def timing_signal(length, channels, min_timescale=1.0, max_timescale=1.0e5, dtype=np.float32):
    """third_party/tensor2tensor/common_attention.py:28-69 (max_timescale 1e5, Q2).
    Returns [length, channels]; the op order follows the TF graph in `dtype`."""
    dt = np.dtype(dtype).type
    f64 = np.float64
    position = np.arange(length).astype(dtype)
    num_timescales = channels // 2
    log_timescale_increment = dt(math.log(float(max_timescale) / float(min_timescale))) / \
        (dt(num_timescales) - dt(1))
    # exp / sin / cos are taken correctly rounded in `dtype` (evaluated in float64, then rounded):
    # TF's own float32 kernels are not reproducible here, this is the ideal float32 evaluation.
    inv_timescales = dt(min_timescale) * np.exp(
        (np.arange(num_timescales).astype(dtype) * -log_timescale_increment).astype(f64)).astype(dtype)
    scaled_time = (position[:, None] * inv_timescales[None, :]).astype(f64)
    signal = np.concatenate([np.sin(scaled_time), np.cos(scaled_time)], axis=1).astype(dtype)
    if channels % 2:
        signal = np.pad(signal, [[0, 0], [0, 1]])
    return signal.astype(dtype)


def dense(x, kernel, bias=None):
    """utils.py:546-557 -- reshape(x,[-1,in]) @ kernel^T (+ bias); kernel is [out, in]."""
    y = x.reshape(-1, x.shape[-1]) @ kernel.T
    if bias is not None:
        y = y + bias
    return y.reshape(x.shape[:-1] + (kernel.shape[0],))


def layer_norm(x, scale, bias, epsilon=1e-6):
    """third_party/tensor2tensor/common_layers.py:333-338 (eps from :356)."""
    dt = x.dtype.type
    mean = x.mean(axis=-1, keepdims=True)
    variance = np.square(x - mean).mean(axis=-1, keepdims=True)
    norm_x = (x - mean) * (dt(1) / np.sqrt(variance + dt(epsilon)))
    return norm_x * scale + bias


def contrib_layer_norm(x, gamma, beta):
    """tf.contrib.layers.layer_norm(x) as called at model.py:453 with TF1 defaults (Q1):
    moments over all non-batch axes (T_padded x H), variance_epsilon 1e-12, gamma/beta [H];
    tf.nn.batch_normalization computes x*inv + (beta - mean*inv), inv = rsqrt(var+eps)*gamma."""
    dt = x.dtype.type
    axes = tuple(range(1, x.ndim))
    mean = x.mean(axis=axes, keepdims=True)
    variance = np.square(x - mean).mean(axis=axes, keepdims=True)
    inv = (dt(1) / np.sqrt(variance + dt(1e-12))) * gamma
    return x * inv + (beta - mean * inv)


def softmax(x):
    m = x.max(axis=-1, keepdims=True)
    e = np.exp(x - m)
    return e / e.sum(axis=-1, keepdims=True)


def log_softmax(x):
    """tf.nn.log_softmax: shifted = x - max; shifted - log(sum(exp(shifted)))."""
    shifted = x - x.max(axis=-1, keepdims=True)
    return shifted - np.log(np.exp(shifted).sum(axis=-1, keepdims=True))


def split_heads(x, n):
    """common_attention.py:243-253  [B, L, C] -> [B, n, L, C/n]."""
    b, l, c = x.shape
    return x.reshape(b, l, n, c // n).transpose(0, 2, 1, 3)


def combine_heads(x):
    """common_attention.py:256-265."""
    b, n, l, d = x.shape
    return x.transpose(0, 2, 1, 3).reshape(b, l, n * d)


def dot_product_attention(q, k, v, bias):
    """common_attention.py:295-334 (dropout is identity at inference)."""
    logits = q @ k.transpose(0, 1, 3, 2)
    if bias is not None:
        logits = logits + bias
    return softmax(logits) @ v


def multihead_attention(W, prefix, query_antecedent, memory_antecedent, bias, num_heads,
                        reserve_last=False):
    """utils.py:571-633."""
    dt = query_antecedent.dtype.type
    if memory_antecedent is None:
        combined = dense(query_antecedent, W[prefix + '/qkv_transform/kernel'],
                         W[prefix + '/qkv_transform/bias'])
        q, k, v = np.split(combined, 3, axis=2)
    else:
        q = dense(query_antecedent, W[prefix + '/q_transform/kernel'], W[prefix + '/q_transform/bias'])
        combined = dense(memory_antecedent, W[prefix + '/kv_transform/kernel'],
                         W[prefix + '/kv_transform/bias'])
        k, v = np.split(combined, 2, axis=2)
    if reserve_last:
        q = q[:, -1:, :]
    q, k, v = split_heads(q, num_heads), split_heads(k, num_heads), split_heads(v, num_heads)
    key_depth_per_head = q.shape[-1]
    q = q * dt(key_depth_per_head ** -0.5)          # Q3: after the bias
    x = combine_heads(dot_product_attention(q, k, v, bias))
    return dense(x, W[prefix + '/output_transform/kernel'], W[prefix + '/output_transform/bias'])


def _sigmoid(x):
    return x.dtype.type(1) / (x.dtype.type(1) + np.exp(-x))


def ff_hidden(W, prefix, x, activation='glu'):
    """utils.py:559-568 with the activations of model.py:437-441."""
    if activation == 'glu':
        a1 = dense(x, W[prefix + '/dense/arg1/kernel'], W[prefix + '/dense/arg1/bias'])
        a2 = dense(x, W[prefix + '/dense/arg2/kernel'], W[prefix + '/dense/arg2/bias'])
        h = a1 * _sigmoid(a2)
    else:
        h = dense(x, W[prefix + '/dense/kernel'], W[prefix + '/dense/bias'])
        if activation == 'relu':
            h = np.maximum(h, 0)
        elif activation == 'sigmoid':
            h = _sigmoid(h)
        elif activation == 'tanh':
            h = np.tanh(h)
        elif activation == 'swish':
            h = h * _sigmoid(h)
        else:
            raise ValueError(activation)
    return dense(h, W[prefix + '/dense_1/kernel'], W[prefix + '/dense_1/bias'])


def residual(W, ln_prefix, inputs, outputs):
    """utils.py:494-507."""
    return layer_norm(inputs + outputs, W[ln_prefix + '/layer_norm_scale'],
                      W[ln_prefix + '/layer_norm_bias'])


# --------------------------------------------------------------------------------------
# model
# --------------------------------------------------------------------------------------
class Config(dict):
    """Minimal attribute dict for the oracle (the product's AttrDict lives in asr_amd)."""
    __getattr__ = dict.get


def make_config(hidden_units=1024, num_heads=16, num_blocks=6, dst_vocab_size=3900, input_dim=320,
                ff_activation='glu', scale_embedding=True, tie_embedding_and_softmax=True,
                beam_size=10, max_target_length=200, lp_alpha=0.6):
    return Config(hidden_units=hidden_units, num_heads=num_heads, num_blocks=num_blocks,
                  dst_vocab_size=dst_vocab_size, input_dim=input_dim, ff_activation=ff_activation,
                  scale_embedding=scale_embedding, tie_embedding_and_softmax=tie_embedding_and_softmax,
                  beam_size=beam_size, max_target_length=max_target_length, lp_alpha=lp_alpha)


def variable_shapes(cfg):
    """SURVEY A.1 -- checkpoint variable names and shapes of the test graph."""
    H, D, V, L = cfg.hidden_units, cfg.input_dim, cfg.dst_vocab_size, cfg.num_blocks
    F = 4 * H
    s = {}
    s['encoder/src_change/kernel'] = (H, D)
    s['encoder/src_change/bias'] = (H,)
    s['encoder/LayerNorm/beta'] = (H,)
    s['encoder/LayerNorm/gamma'] = (H,)

    def ffn(p):
        if cfg.ff_activation == 'glu':
            for a in ('arg1', 'arg2'):
                s[p + '/ff_hidden/dense/%s/kernel' % a] = (F, H)
                s[p + '/ff_hidden/dense/%s/bias' % a] = (F,)
        else:
            s[p + '/ff_hidden/dense/kernel'] = (F, H)
            s[p + '/ff_hidden/dense/bias'] = (F,)
        s[p + '/ff_hidden/dense_1/kernel'] = (H, F)
        s[p + '/ff_hidden/dense_1/bias'] = (H,)

    def ln(p, names):
        for n in names:
            s[p + '/%s/layer_norm_scale' % n] = (H,)
            s[p + '/%s/layer_norm_bias' % n] = (H,)

    for i in range(L):
        p = 'encoder/block_%d' % i
        s[p + '/encoder_self_attention/qkv_transform/kernel'] = (3 * H, H)
        s[p + '/encoder_self_attention/qkv_transform/bias'] = (3 * H,)
        s[p + '/encoder_self_attention/output_transform/kernel'] = (H, H)
        s[p + '/encoder_self_attention/output_transform/bias'] = (H,)
        ffn(p)
        ln(p, ('layer_norm', 'layer_norm_1'))
    s['decoder/dst_embedding/kernel'] = (V, H)
    if not cfg.tie_embedding_and_softmax:
        s['decoder/softmax/kernel'] = (V, H)
    for i in range(L):
        p = 'decoder/block_%d' % i
        s[p + '/decoder_self_attention/qkv_transform/kernel'] = (3 * H, H)
        s[p + '/decoder_self_attention/qkv_transform/bias'] = (3 * H,)
        s[p + '/decoder_self_attention/output_transform/kernel'] = (H, H)
        s[p + '/decoder_self_attention/output_transform/bias'] = (H,)
        s[p + '/decoder_vanilla_attention/q_transform/kernel'] = (H, H)
        s[p + '/decoder_vanilla_attention/q_transform/bias'] = (H,)
        s[p + '/decoder_vanilla_attention/kv_transform/kernel'] = (2 * H, H)
        s[p + '/decoder_vanilla_attention/kv_transform/bias'] = (2 * H,)
        s[p + '/decoder_vanilla_attention/output_transform/kernel'] = (H, H)
        s[p + '/decoder_vanilla_attention/output_transform/bias'] = (H,)
        ffn(p)
        ln(p, ('layer_norm', 'layer_norm_1', 'layer_norm_2'))
    return s


def random_weights(cfg, seed=0, bias_scale=0.0, emb_scale=1.0):
    """Q11: glorot-uniform kernels, zero biases / LN biases, unit LN scales.  `bias_scale` > 0
    draws the biases and LN parameters from small uniforms instead, so that tests exercise the
    bias paths; `emb_scale` rescales the (tied) embedding to sharpen the output posteriors."""
    rng = np.random.default_rng(seed)
    W = {}
    for name, shape in sorted(variable_shapes(cfg).items()):
        if name.endswith('/kernel'):
            limit = math.sqrt(6.0 / (shape[0] + shape[1]))
            w = rng.uniform(-limit, limit, size=shape)
            if name == 'decoder/dst_embedding/kernel' or name == 'decoder/softmax/kernel':
                w = w * emb_scale
        elif name.endswith('gamma') or name.endswith('layer_norm_scale'):
            w = np.ones(shape) + bias_scale * rng.uniform(-1, 1, size=shape)
        else:
            w = bias_scale * rng.uniform(-1, 1, size=shape)
        W[name] = w.astype(np.float32)
    return W


def cast_weights(W, dtype):
    return {k: v.astype(dtype) for k, v in W.items()}


def encoder(X, W, cfg):
    """model.py:444-490.  X [B, T, D] -> [B, T, H]; returns (encoder_output, encoder_padding)."""
    dt = X.dtype.type
    H, nh = cfg.hidden_units, cfg.num_heads
    encoder_padding = (np.abs(X).sum(axis=-1) == 0.0)                       # :450
    x = dense(X, W['encoder/src_change/kernel'], W['encoder/src_change/bias'])   # :451
    x = contrib_layer_norm(x, W['encoder/LayerNorm/gamma'], W['encoder/LayerNorm/beta'])  # :453
    x = x + timing_signal(x.shape[1], H, dtype=X.dtype)[None]               # :456
    bias = (encoder_padding.astype(X.dtype) * dt(MASK_BIAS))[:, None, None, :]   # :470
    for i in range(cfg.num_blocks):
        p = 'encoder/block_%d' % i
        x = residual(W, p + '/layer_norm', x,
                     multihead_attention(W, p + '/encoder_self_attention', x, None, bias, nh))
        x = residual(W, p + '/layer_norm_1', x, ff_hidden(W, p + '/ff_hidden', x, cfg.ff_activation))
    x = x * (dt(1) - encoder_padding.astype(X.dtype))[..., None]            # :489
    return x, encoder_padding


def embed(tokens, W, cfg, dtype):
    """utils.py:523-531 with the multiplier of model.py:572."""
    E = W['decoder/dst_embedding/kernel']
    out = E[tokens]
    if cfg.scale_embedding:
        mult = np.dtype(dtype).type(cfg.hidden_units ** 0.5)
        if mult != 1.0:
            out = out * mult
    return out


def decoder_with_caching(preds, cache, enc, W, cfg):
    """model.py:559-633, literally: block-OUTPUT cache [R, i-1, L, H]; K/V of the whole target
    history and of the whole (beam-tiled) encoder memory are re-projected on every call."""
    dt = enc.dtype.type
    nh = cfg.num_heads
    enc_pad = (np.abs(enc).sum(axis=-1) == 0.0)                             # :566
    enc_bias = (enc_pad.astype(enc.dtype) * dt(MASK_BIAS))[:, None, None, :]
    y = embed(preds, W, cfg, enc.dtype)                                     # :569-573
    y = y + timing_signal(y.shape[1], cfg.hidden_units, dtype=enc.dtype)[None]   # :576
    new_cache = []
    for i in range(cfg.num_blocks):
        p = 'decoder/block_%d' % i
        z = residual(W, p + '/layer_norm', y[:, -1:, :],
                     multihead_attention(W, p + '/decoder_self_attention', y, None, None, nh,
                                         reserve_last=True))               # :588-601
        z = residual(W, p + '/layer_norm_1', z,
                     multihead_attention(W, p + '/decoder_vanilla_attention', z, enc, enc_bias, nh,
                                         reserve_last=True))               # :604-617
        z = residual(W, p + '/layer_norm_2', z, ff_hidden(W, p + '/ff_hidden', z, cfg.ff_activation))
        y = np.concatenate([cache[:, :, i, :], z], axis=1)                  # :628
        new_cache.append(y[:, :, None, :])
    return y, np.concatenate(new_cache, axis=2)                             # :631-633


class KVState(object):
    """Decoder state of the K/V-cache formulation (SURVEY A.2 'KV-cache equivalence')."""

    def __init__(self, enc, W, cfg, beam_size):
        dt = enc.dtype
        B, T, H = enc.shape
        L, nh = cfg.num_blocks, cfg.num_heads
        self.cfg, self.W, self.K = cfg, W, beam_size
        enc_pad = (np.abs(enc).sum(axis=-1) == 0.0)
        self.enc_bias = (enc_pad.astype(dt) * dt.type(MASK_BIAS))[:, None, None, :]   # [B,1,1,T]
        self.cross_k, self.cross_v = [], []
        for l in range(L):
            p = 'decoder/block_%d/decoder_vanilla_attention' % l
            kv = dense(enc, W[p + '/kv_transform/kernel'], W[p + '/kv_transform/bias'])
            k, v = np.split(kv, 2, axis=2)
            self.cross_k.append(split_heads(k, nh))       # [B, nh, T, d]
            self.cross_v.append(split_heads(v, nh))
        R = B * beam_size
        self.self_k = [np.zeros((R, nh, 0, H // nh), dt) for _ in range(L)]
        self.self_v = [np.zeros((R, nh, 0, H // nh), dt) for _ in range(L)]
        self.pe = None

    def reorder(self, parent):
        self.self_k = [k[parent] for k in self.self_k]
        self.self_v = [v[parent] for v in self.self_v]

    def step(self, tokens, pos):
        """tokens [R] at decoder position `pos` -> last-block output [R, H]."""
        cfg, W, K = self.cfg, self.W, self.K
        nh, H = cfg.num_heads, cfg.hidden_units
        R = tokens.shape[0]
        B = R // K
        dtp = self.cross_k[0].dtype
        dt = dtp.type
        d = H // nh
        if self.pe is None or self.pe.shape[0] <= pos:
            self.pe = timing_signal(max(2 * (pos + 1), 64), H, dtype=dtp)
        y = embed(tokens, W, cfg, dtp) + self.pe[pos]                        # [R, H]
        for l in range(cfg.num_blocks):
            p = 'decoder/block_%d' % l
            sa = p + '/decoder_self_attention'
            qkv = y @ W[sa + '/qkv_transform/kernel'].T + W[sa + '/qkv_transform/bias']
            q, k, v = np.split(qkv, 3, axis=1)
            self.self_k[l] = np.concatenate([self.self_k[l], k.reshape(R, nh, 1, d)], axis=2)
            self.self_v[l] = np.concatenate([self.self_v[l], v.reshape(R, nh, 1, d)], axis=2)
            q = q.reshape(R, nh, 1, d) * dt(d ** -0.5)
            a = softmax(q @ self.self_k[l].transpose(0, 1, 3, 2)) @ self.self_v[l]   # [R,nh,1,d]
            a = a.reshape(R, H) @ W[sa + '/output_transform/kernel'].T + W[sa + '/output_transform/bias']
            z = layer_norm(y + a, W[p + '/layer_norm/layer_norm_scale'], W[p + '/layer_norm/layer_norm_bias'])
            ca = p + '/decoder_vanilla_attention'
            q = (z @ W[ca + '/q_transform/kernel'].T + W[ca + '/q_transform/bias']) * dt(d ** -0.5)
            q = q.reshape(B, K, nh, d).transpose(0, 2, 1, 3)                 # [B, nh, K, d]
            logits = q @ self.cross_k[l].transpose(0, 1, 3, 2) + self.enc_bias    # [B, nh, K, T]
            a = softmax(logits) @ self.cross_v[l]                            # [B, nh, K, d]
            a = a.transpose(0, 2, 1, 3).reshape(R, H)
            a = a @ W[ca + '/output_transform/kernel'].T + W[ca + '/output_transform/bias']
            z = layer_norm(z + a, W[p + '/layer_norm_1/layer_norm_scale'], W[p + '/layer_norm_1/layer_norm_bias'])
            f = ff_hidden(W, p + '/ff_hidden', z, cfg.ff_activation)
            y = layer_norm(z + f, W[p + '/layer_norm_2/layer_norm_scale'], W[p + '/layer_norm_2/layer_norm_bias'])
        return y


def output_logits(h_last, W, cfg):
    """model.py:341-343 -- tied embedding (or decoder/softmax), no bias."""
    name = 'decoder/dst_embedding/kernel' if cfg.tie_embedding_and_softmax else 'decoder/softmax/kernel'
    return h_last @ W[name].T


def top_k(z, k):
    """tf.nn.top_k restated as: value descending, ties -> lower index (Q5 for sorted=False:
    the candidate ORDER is implementation-defined in TF; this choice is the build's)."""
    idx = np.argsort(-z, axis=-1, kind='stable')[..., :k]
    return np.take_along_axis(z, idx, axis=-1), idx.astype(np.int32)


def beam_bookkeeping(scores, fin, cand_scores, cand_tokens, K):
    """One pruning step, model.py:280-301.  scores [R], fin [R] bool,
    cand_* [R, K] -> (scores', tokens', parents', fin')."""
    dtp = scores.dtype
    dt = dtp.type
    R = scores.shape[0]
    B = R // K
    f = fin.astype(dtp)
    fi = fin.astype(np.int32)
    cand_tokens = cand_tokens * (1 - fi[:, None]) + fi[:, None] * 3                       # :253
    b = np.array([0.0] + [-NEG_INF_BEAM] * (K - 1), dtype=dtp)[None, :]
    cand_scores = cand_scores * (dt(1) - f[:, None]) + b * f[:, None]                      # :240
    tot = (scores[:, None] + cand_scores).reshape(B, K * K)                                # :284-285
    new_scores, k_idx = top_k(tot, K)                                                      # :288
    g = (np.arange(B, dtype=np.int64)[:, None] * (K * K) + k_idx).reshape(-1)              # :290-292
    tokens = cand_tokens.reshape(-1)[g].astype(np.int32)                                   # :295
    parents = (g // K).astype(np.int32)                                                    # :296 (Py2 floor div)
    return new_scores.reshape(-1), tokens, parents, tokens == 3                            # :301


def finalize(preds, scores, B, K, lp_alpha):
    """model.py:325-336 -- GNMT length penalty (length counts the leading <S>, Q8), argmax."""
    dtp = scores.dtype
    dt = dtp.type
    p3 = preds.reshape(B, K, -1)
    lengths = (p3 != 3).astype(dtp).sum(axis=-1)
    lp = np.power((dt(5) + lengths) / dt(6), dt(lp_alpha))
    best = np.argmax(scores.reshape(B, K) / lp, axis=-1)
    return p3[np.arange(B), best][:, 1:].astype(np.int32)


def beam_search(enc, W, cfg, algo='kv', trace=None):
    """model.py:221-336.  enc [B, T, H] (one shard) -> int32 [B, steps].
    algo='ref' runs decoder_with_caching literally; algo='kv' the K/V-cache formulation.
    `trace`, if a dict, receives per-step 'logits', 'scores', 'tokens', 'parents'."""
    B, T, H = enc.shape
    K = cfg.beam_size
    R = B * K
    dtp = enc.dtype
    if B == 0:
        return np.zeros((0, 0), np.int32)                                     # model.py:184-185
    preds = np.full((R, 1), BOS_INDEX, np.int32)                             # :263
    scores = np.tile(np.array([0.0] + [-NEG_INF_BEAM] * (K - 1), dtype=dtp), B)   # :264-265
    fin = np.zeros(R, bool)                                                  # :266
    limit = min(T + 50, cfg.max_target_length)                               # :310
    i = 0
    if algo == 'ref':
        enc_tiled = np.repeat(enc, K, axis=0)                                # :257-261
        cache = np.zeros((R, 0, cfg.num_blocks, H), dtp)                     # :268
    else:
        state = KVState(enc, W, cfg, K)
    while np.any(~fin) and i <= limit:                                       # :305-312
        i += 1
        if algo == 'ref':
            dec_out, cache = decoder_with_caching(preds, cache, enc_tiled, W, cfg)
            h_last = dec_out[:, -1]
        else:
            h_last = state.step(preds[:, -1], i - 1)
        logits = output_logits(h_last, W, cfg)                               # :341
        z = log_softmax(logits)                                              # :345
        cs, ct = top_k(z, K)                                                 # :346
        scores, tokens, parents, fin = beam_bookkeeping(scores, fin, cs, ct, K)
        preds = np.concatenate([preds[parents], tokens[:, None]], axis=1)    # :296,298
        if algo == 'ref':
            cache = cache[parents]                                           # :297
        else:
            state.reorder(parents)
        if trace is not None:
            trace.setdefault('logits', []).append(logits)
            trace.setdefault('scores', []).append(scores.copy())
            trace.setdefault('tokens', []).append(tokens.copy())
            trace.setdefault('parents', []).append(parents.copy())
    if trace is not None:
        trace['preds'] = preds
        trace['final_scores'] = scores
    return finalize(preds, scores, B, K, cfg.lp_alpha)


def expand_shards(n_rows, n):
    """utils.py:430-442 -- dim-0 shard sizes: the first (n_rows % n) shards get one extra."""
    span, rem = n_rows // n, n_rows % n
    out, base = [], 0
    for i in range(n):
        end = base + span + (1 if i < rem else 0)
        out.append((base, end))
        base = end
    return out


def build_test_model(X, W, cfg, num_gpus=1, algo='kv', dtype=np.float32):
    """model.py:162-204 + evaluate.py:46-47: shard dim 0 over `num_gpus` towers, run
    encoder -> beam_search per shard, pad with 3 to the longest, concat (Q10)."""
    X = np.asarray(X, dtype=dtype)
    Wc = cast_weights(W, dtype)
    preds = []
    for (a, b) in expand_shards(X.shape[0], max(num_gpus, 1)):
        if b - a == 0:
            preds.append(np.zeros((0, 0), np.int32))
            continue
        enc, _ = encoder(X[a:b], Wc, cfg)
        preds.append(beam_search(enc, Wc, cfg, algo=algo))
    max_len = max(p.shape[1] for p in preds)
    preds = [np.concatenate([p, np.full((p.shape[0], max_len - p.shape[1]), 3, np.int32)], axis=1)
             for p in preds]
    return np.concatenate(preds, axis=0)


def shift_right(Y, pad=BOS_INDEX):
    """utils.py:518-520: decoder input = <S> followed by the targets but the last."""
    Y = np.asarray(Y)
    return np.concatenate([np.full_like(Y[:, :1], pad), Y[:, :-1]], axis=1)


def decoder(dec_input, enc, W, cfg):
    """model.py:492-557 (decoder_impl, teacher forced): causal self-attention with the additive -1e9 upper
    triangle of attention_bias_lower_triangle (common_attention.py:179-190), cross-attention with the
    re-detected encoder padding (:497-498)."""
    dt = enc.dtype.type
    nh = cfg.num_heads
    L = dec_input.shape[1]
    enc_pad = (np.abs(enc).sum(axis=-1) == 0.0)                             # :497
    enc_bias = (enc_pad.astype(enc.dtype) * dt(MASK_BIAS))[:, None, None, :]
    y = embed(dec_input, W, cfg, enc.dtype)                                 # :500-504
    y = y + timing_signal(L, cfg.hidden_units, dtype=enc.dtype)[None]       # :507
    lower = np.tril(np.ones((L, L), enc.dtype))
    self_bias = (dt(MASK_BIAS) * (dt(1) - lower))[None, None]               # :513
    for i in range(cfg.num_blocks):
        p = 'decoder/block_%d' % i
        y = residual(W, p + '/layer_norm', y,
                     multihead_attention(W, p + '/decoder_self_attention', y, None, self_bias, nh))     # :516-529
        y = residual(W, p + '/layer_norm_1', y,
                     multihead_attention(W, p + '/decoder_vanilla_attention', y, enc, enc_bias, nh))    # :532-544
        y = residual(W, p + '/layer_norm_2', y, ff_hidden(W, p + '/ff_hidden', y, cfg.ff_activation))   # :547-555
    return y


def test_loss(dec_out, Y, W, cfg):
    """model.py:350-360: sum over positions with Y != 0 of softmax_cross_entropy_with_logits(logits, one_hot(Y))."""
    logits = output_logits(dec_out, W, cfg)
    z = log_softmax(logits)
    Y = np.asarray(Y)
    V = cfg.dst_vocab_size
    valid = (Y >= 0) & (Y < V)                                              # one_hot of an out-of-range id is all zeros
    picked = np.take_along_axis(z, np.clip(Y, 0, V - 1)[..., None], axis=-1)[..., 0]
    loss = np.where(valid, -picked, 0.0)
    return float(np.sum(loss.astype(np.float64) * (Y != 0)))


test_loss.__test__ = False        # not a pytest test


def loss_sum(X, Y, W, cfg, num_gpus=1, dtype=np.float32):
    """evaluate.py:49-50 -> model.loss_sum (model.py:168-190): towers as in build_test_model, losses added."""
    X = np.asarray(X, dtype=dtype)
    Y = np.asarray(Y, dtype=np.int32)
    Wc = cast_weights(W, dtype)
    total = 0.0
    for (a, b) in expand_shards(X.shape[0], max(num_gpus, 1)):
        if b - a == 0:
            continue                                                         # false_fn: 0.0 (:184-187)
        enc, _ = encoder(X[a:b], Wc, cfg)
        total += test_loss(decoder(shift_right(Y[a:b]), enc, Wc, cfg), Y[a:b], Wc, cfg)
    return total


def indices_to_words(Y, idx2word):
    """utils.py:405-417."""
    sents = []
    for y in Y:
        sent = []
        for i in y:
            if i == 3:
                break
            sent.append(idx2word[int(i)])
        sents.append(' '.join(sent))
    return sents


# --------------------------------------------------------------------------------------
# synthetic inputs
# --------------------------------------------------------------------------------------
def synthetic_fbank(n_utts, t_raw, seed=0, dim=80):
    """Raw fbank ~ N(0,1), seeded; `t_raw` is an int or a per-utterance sequence."""
    rng = np.random.default_rng(seed)
    lens = [t_raw] * n_utts if np.isscalar(t_raw) else list(t_raw)
    return [rng.standard_normal((int(t), dim)).astype(np.float32) for t in lens]


def make_batch(raw_list, left=3, right=0, sub=3):
    """frame_stack each utterance, then zero-pad to the batch maximum (utils.py:204-220)."""
    feats = [frame_stack(r, left, right, sub) for r in raw_list]
    tmax = max(f.shape[0] for f in feats)
    X = np.zeros((len(feats), tmax, feats[0].shape[1]), np.float32)
    for i, f in enumerate(feats):
        X[i, :f.shape[0]] = f
    return X
