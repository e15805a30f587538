"""The allocation-free entry point asr_op_input_norm_timing_noalloc (include/asr_amd.h; no counterpart in the
reference: for callers that cannot hipMalloc) against the existing asr_op_input_norm_timing, through the C ABI.

Same kernels, same order of operations: the outputs must be BIT-identical.  Two calls in a row on one handle, the
second with a smaller input (fewer utterances, fewer frames): the static scratch keeps the first call's contents and
nothing may leak from them.  Collected last (file name) so that a failure here cannot hide another test."""
import ctypes as C

import numpy as np
import pytest

from oracle import asr_oracle as O
from tests.helpers import product_config, tiny_config

pytestmark = pytest.mark.gpu


def _existing(eng, x, gamma, beta):
    from asr_amd import _lib
    n, t, hdim = x.shape
    y = np.empty_like(x)
    _lib.check(eng._h, _lib.lib.asr_op_input_norm_timing(eng._h, _lib.fptr(x), _lib.fptr(gamma), _lib.fptr(beta), n, t, hdim,
                                                         _lib.fptr(y)))
    return y


def _noalloc(eng, dx, dg, db, dy):
    from asr_amd import _lib
    n, t, hdim = dx.shape
    return _lib.lib.asr_op_input_norm_timing_noalloc(eng._h, C.c_void_p(dx.data_ptr()), C.c_void_p(dg.data_ptr()),
                                                     C.c_void_p(db.data_ptr()), n, t, hdim, C.c_void_p(dy.data_ptr()))


def test_noalloc_entry_point_matches_the_allocating_one_twice_in_a_row():
    import torch
    from asr_amd import _lib
    from asr_amd.model import Engine
    ocfg = tiny_config()                                           # hidden 128; timing table: max(512, mtl + 2) positions
    eng = Engine(product_config(ocfg), 0)
    try:
        rng = np.random.default_rng(5)
        gamma = (1.0 + 0.1 * rng.standard_normal(128)).astype(np.float32)
        beta = (0.1 * rng.standard_normal(128)).astype(np.float32)
        dg, db = torch.from_numpy(gamma).cuda(), torch.from_numpy(beta).cuda()
        for (n, t) in ((7, 301), (3, 40), (1, 1), (64, 334)):      # the second and third calls are smaller than the first
            x = (rng.standard_normal((n, t, 128)) * 3.0 + 0.5).astype(np.float32)
            want = _existing(eng, x, gamma, beta)
            # float64 oracle of the same op, so the comparison is not only self-consistency
            ref = O.contrib_layer_norm(x.astype(np.float64), gamma.astype(np.float64), beta.astype(np.float64)) + \
                O.timing_signal(t, 128)[None]
            assert np.max(np.abs(want - ref)) < 5e-5
            dx = torch.from_numpy(x).cuda()
            dy = torch.full((n, t, 128), float('nan'), dtype=torch.float32, device='cuda')
            torch.cuda.synchronize()
            mem_before = torch.cuda.mem_get_info()[0]
            rc = _noalloc(eng, dx, dg, db, dy)
            _lib.check(eng._h, rc)
            assert torch.cuda.mem_get_info()[0] == mem_before      # free device memory unchanged by the call
            got = dy.cpu().numpy()
            assert np.array_equal(got, want), (n, t)
        # limits are rejected with a clear error and without touching the output
        dx = torch.zeros((1025, 1, 128), dtype=torch.float32, device='cuda')
        dy = torch.full((1025, 1, 128), 7.0, dtype=torch.float32, device='cuda')
        torch.cuda.synchronize()
        rc = _noalloc(eng, dx, dg, db, dy)
        assert rc == 5 and b'1024' in _lib.lib.asr_last_error(eng._h)             # ASR_ERR_CAPACITY
        assert bool((dy == 7.0).all())
        dx = torch.zeros((1, 513, 128), dtype=torch.float32, device='cuda')
        dy = torch.zeros((1, 513, 128), dtype=torch.float32, device='cuda')
        torch.cuda.synchronize()
        rc = _noalloc(eng, dx, dg, db, dy)
        assert rc == 5 and b'timing table' in _lib.lib.asr_last_error(eng._h)
        # a host pointer is refused instead of faulting the GPU
        xh = np.zeros((1, 4, 128), np.float32)
        rc = _lib.lib.asr_op_input_norm_timing_noalloc(eng._h, C.c_void_p(xh.ctypes.data), C.c_void_p(dg.data_ptr()),
                                                       C.c_void_p(db.data_ptr()), 1, 4, 128, C.c_void_p(dy.data_ptr()))
        assert rc == 1
        # and the handle still works
        x = rng.standard_normal((2, 9, 128)).astype(np.float32)
        dx = torch.from_numpy(x).cuda()
        dy = torch.empty((2, 9, 128), dtype=torch.float32, device='cuda')
        torch.cuda.synchronize()
        _lib.check(eng._h, _noalloc(eng, dx, dg, db, dy))
        assert np.array_equal(dy.cpu().numpy(), _existing(eng, x, gamma, beta))
    finally:
        eng.close()
