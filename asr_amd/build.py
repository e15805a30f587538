"""Build libasr_amd.so (HIP, gfx950 only) in-tree: `python -m asr_amd.build`.

hipcc cross-compiles without a GPU, so this runs in the CPU-only build container; the resulting
asr_amd/csrc/libasr_amd.so travels to the GPU box with the tree.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'csrc')
LIB = os.path.join(CSRC, 'libasr_amd.so')
SOURCES = ['gemm_f32.hip', 'gemm_split.hip', 'rowops.hip', 'frontend.hip', 'attention.hip', 'engine.hip']
HEADERS = ['asr_common.h', 'gemm_epilogue.h', os.path.join('..', '..', 'include', 'asr_amd.h')]
FLAGS = ['-O3', '-std=c++17', '--offload-arch=gfx950', '-fPIC', '-Wall', '-Wno-unused-function']


def _hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError('hipcc not found')


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=True):
    hipcc = _hipcc()
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs, jobs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace('.hip', '.o'))
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            jobs.append([hipcc] + FLAGS + ['-c', s, '-o', o])

    def run(cmd):
        if verbose:
            print(' '.join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed:\n' + r.stdout + r.stderr)
        return r

    with ThreadPoolExecutor(max_workers=min(4, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(LIB, objs):
        run([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs)
    return LIB


if __name__ == '__main__':
    print(build_library(force='--force' in sys.argv))
