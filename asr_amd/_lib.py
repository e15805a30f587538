"""ctypes binding of libasr_amd.so (the C ABI declared in include/asr_amd.h).

There is no CPU fallback: if the HIP library has not been built, importing this module raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# ASR_AMD_LIB: development switch (A/B timing of two builds on one GPU box); the product loads the in-tree library
LIB_PATH = os.environ.get('ASR_AMD_LIB') or os.path.join(_HERE, 'csrc', 'libasr_amd.so')

ASR_OK = 0
FF_ACTIVATIONS = {'relu': 0, 'glu': 1, 'sigmoid': 2, 'tanh': 3, 'swish': 4}
PRECISIONS = {'f32': 0, 'f32_bf16x6': 1, 'bf16x3': 2, 'bf16': 3, 'f32_f16x3': 4}


class AsrError(RuntimeError):
    def __init__(self, code, message):
        super(AsrError, self).__init__('libasr_amd error %d: %s' % (code, message))
        self.code = code


class asr_config(C.Structure):
    _fields_ = [('input_dim', C.c_int32), ('hidden_units', C.c_int32), ('num_heads', C.c_int32),
                ('num_blocks', C.c_int32), ('dst_vocab_size', C.c_int32), ('ff_activation', C.c_int32),
                ('scale_embedding', C.c_int32), ('tie_embedding_and_softmax', C.c_int32),
                ('beam_size', C.c_int32), ('max_target_length', C.c_int32), ('lp_alpha', C.c_float),
                ('device', C.c_int32), ('precision', C.c_int32), ('sub_batches', C.c_int32),
                ('shared_gpu', C.c_int32), ('reserved', C.c_int32 * 1)]


def _preload_torch_hip_runtime():
    """One process can host only one HIP/HSA runtime.  The PyTorch ROCm wheel bundles its own
    libamdhip64.so (soname libamdhip64.so.7, the same soname libasr_amd.so needs), so when PyTorch is
    installed bind to that copy up front: whichever of torch / asr_amd is imported first, both then
    share one runtime (bench.py uses torch.distributed/RCCL next to this library).  Without PyTorch
    the library's RUNPATH (/opt/rocm) is used."""
    if os.environ.get('ASR_AMD_SYSTEM_HIP', '') == '1':
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec('torch')
        if spec is None or not spec.submodule_search_locations:
            return
        path = os.path.join(list(spec.submodule_search_locations)[0], 'lib', 'libamdhip64.so')
        if os.path.exists(path):
            C.CDLL(path, mode=C.RTLD_GLOBAL)
    except (OSError, ImportError, ValueError):
        pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            'asr_amd: %s is missing -- build it with `python -m asr_amd.build` (needs hipcc, '
            'cross-compiles for gfx950 without a GPU). There is no CPU fallback.' % LIB_PATH)
    _preload_torch_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64, f32p, i32p, i64p = C.c_void_p, C.c_int32, C.c_int64, C.POINTER(C.c_float), \
        C.POINTER(C.c_int32), C.POINTER(C.c_int64)
    sig = {
        'asr_device_count': (i32, []),
        'asr_create': (i32, [C.POINTER(asr_config), C.POINTER(vp)]),
        'asr_destroy': (None, [vp]),
        'asr_last_error': (C.c_char_p, [vp]),
        'asr_load_tensor': (i32, [vp, C.c_char_p, f32p, i64p, i32]),
        'asr_finalize_weights': (i32, [vp]),
        'asr_share_weights': (i32, [vp, vp]),
        'asr_set_sharing': (i32, [vp, i32, i32]),
        'asr_decode': (i32, [vp, f32p, i32, i32, i32, i32p, i64, i32p]),
        'asr_decode_dev': (i32, [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32p, i64, i32p]),
        'asr_loss': (i32, [vp, f32p, i32, i32, i32, i32p, i32, C.POINTER(C.c_double)]),
        'asr_stack_frames': (i32, [vp, f32p, i32, i32, i32, i32, i32, f32p]),
        'asr_fea_add_delt': (i32, [vp, f32p, i32, i32, f32p]),
        'asr_fea_cmvn_global': (i32, [vp, f32p, i32, i32, f32p, f32p, f32p]),
        'asr_fea_cmvn_utt': (i32, [vp, f32p, i32, i32, f32p]),
        'asr_fea_splice': (i32, [vp, f32p, i32, i32, i32, i32, i32, i32, i32, f32p]),
        'asr_encode': (i32, [vp, f32p, i32, i32, i32, f32p, i32p]),
        'asr_decode_trace': (i32, [vp, f32p, i32, i32, i32, i32p, i64, i32p, i32, f32p, f32p, i32p, i32p]),
        'asr_op_dense': (i32, [vp, f32p, f32p, f32p, i32, i32, i32, i32, f32p]),
        'asr_op_dense_glu': (i32, [vp, f32p, f32p, f32p, f32p, f32p, i32, i32, i32, f32p]),
        'asr_op_add_layer_norm': (i32, [vp, f32p, f32p, f32p, f32p, f32p, i32, i32, f32p]),
        'asr_op_input_norm_timing': (i32, [vp, f32p, f32p, f32p, i32, i32, i32, f32p]),
        'asr_op_embedding': (i32, [vp, f32p, i32p, i32, i32, i32, i32, i32, f32p]),
        'asr_op_pack_cross_kv': (i32, [vp, f32p, i32, i32, i32, i32, vp, vp]),
        'asr_decode_groups': (i32, [vp, f32p, i32, i32p, i32p, i32, i32p, i64, i32p]),
        'asr_op_decoder_self_attention': (i32, [vp, f32p, f32p, f32p, f32p, i32p, i32, i32, i32, i32, i32, f32p]),
        'asr_op_decoder_cross_attention': (i32, [vp, f32p, f32p, f32p, i32p, i32, i32, i32, i32, i32, f32p, i32p]),
        'asr_op_input_norm_timing_noalloc': (i32, [vp, vp, vp, vp, i32, i32, i32, vp]),
        'asr_op_encoder_attention': (i32, [vp, f32p, i32p, i32, i32, f32p]),
        'asr_op_log_softmax_topk': (i32, [vp, f32p, i32, i32, i32, f32p, i32p]),
        'asr_op_beam_step': (i32, [vp, f32p, i32p, f32p, i32p, i32, i32, f32p, i32p, i32p, i32p]),
        'asr_set_profile': (i32, [vp, i32]),
        'asr_get_profile': (i32, [vp, i32, C.POINTER(C.c_char_p), C.POINTER(C.c_double), i64p,
                                  C.POINTER(C.c_double), C.POINTER(C.c_double), i32p]),
        'asr_device_bytes': (i64, [vp]),
        'asr_range_retries': (i64, [vp]),
        'asr_crc32c': (C.c_uint32, [vp, i64, C.c_uint32]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def fptr(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_float))


def iptr(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_int32))


def check(handle, rc):
    if rc != ASR_OK:
        msg = lib.asr_last_error(handle)
        raise AsrError(rc, msg.decode('utf-8', 'replace') if msg else '')
