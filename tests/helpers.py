"""Shared test helpers: configs in both vocabularies (oracle Config / product AttrDict)."""
import numpy as np

from oracle import asr_oracle as O


def product_config(ocfg, num_gpus=1, precision=None, sub_batches=None):
    """oracle Config -> the AttrDict shape the product reads from the YAML files."""
    from asr_amd.utils import AttrDict
    test_extra = {'precision': precision} if precision else {}       # build extension key (include/asr_amd.h asr_precision)
    if sub_batches:
        test_extra['sub_batches'] = sub_batches                      # build extension key (asr_config.sub_batches)
    return AttrDict({
        'model': 'Transformer',
        'dst_vocab_size': ocfg.dst_vocab_size,
        'hidden_units': ocfg.hidden_units,
        'scale_embedding': ocfg.scale_embedding,
        'tie_embedding_and_softmax': ocfg.tie_embedding_and_softmax,
        'num_blocks': ocfg.num_blocks,
        'num_heads': ocfg.num_heads,
        'ff_activation': ocfg.ff_activation,
        'train': {'input_dim': ocfg.input_dim},
        'test': {'batch_size': 8, 'max_target_length': ocfg.max_target_length, 'lp_alpha': ocfg.lp_alpha,
                 'beam_size': ocfg.beam_size, 'num_gpus': num_gpus, **test_extra},
    })


def tiny_config(**kw):
    base = dict(hidden_units=128, num_heads=2, num_blocks=2, dst_vocab_size=50, input_dim=320,
                beam_size=3, max_target_length=20, tie_embedding_and_softmax=False)
    base.update(kw)
    return O.make_config(**base)


def tiny_inputs(lens=(60, 90, 75), seed=2):
    return O.make_batch(O.synthetic_fbank(len(lens), list(lens), seed=seed))


def rel_err(a, b):
    """How the tests read north_star's "logits within 1e-3 relative error": the largest absolute difference over the
    WHOLE tensor divided by the largest magnitude of the reference tensor (a per-element ratio is meaningless where a
    logit passes through zero).  rel_err_rows below is the stricter per-row (per-hypothesis) form."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-30))


def rel_err_rows(a, b):
    """Per-row variant: max over rows (last axis = one logit vector) of max|a - b| / max|b| of that row."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    a = a.reshape(-1, a.shape[-1])
    b = b.reshape(-1, b.shape[-1])
    den = np.maximum(np.max(np.abs(b), axis=1), 1e-30)
    return float(np.max(np.max(np.abs(a - b), axis=1) / den))
