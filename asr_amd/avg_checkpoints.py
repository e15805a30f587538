"""Checkpoint averaging without TensorFlow: `python -m asr_amd.avg_checkpoints --checkpoints a,b,c --prefix dir/
--output_path dir/averaged.ckpt`.

Mirror of third_party/tensor2tensor/avg_checkpoints.py (README.md:7 "train -> avg_checkpoints -> evaluate"):
  * variables are listed from the FIRST checkpoint, names starting with `global_step` are skipped (:75-79);
  * every other variable - Adam slots and beta powers included - is summed in float64 zeros(shape) and divided by
    the number of checkpoints (:79-88);
  * the result is saved with a fresh int64 `global_step` = 0 as `<output_path>-0` and the directory's `checkpoint`
    state file points at it (:96-107), so `latest_checkpoint(model_dir)` finds `averaged.ckpt-0` (evaluate.py:36).
The reference declares every output variable with the dtype of the LAST variable it read (`var_dtypes[name]` after
the loop, :91 - a quirk that is harmless there because all of them are float32); here each variable keeps its own
floating dtype.  Inputs may be TF-V2 bundles or this build's `.npz` checkpoints; output format `tf` (bundle), `npz`
or `both`.
"""
from __future__ import annotations

import argparse
import os

import numpy as np

from . import tf_bundle


def checkpoint_exists(path):
    return os.path.exists(path) or os.path.exists(path + '.index') or os.path.exists(path + '.npz') or \
        os.path.exists(path + '.meta')


def _read_all(path):
    if path.endswith('.npz') or os.path.exists(path + '.npz'):
        f = path if path.endswith('.npz') else path + '.npz'
        with np.load(f, allow_pickle=False) as z:
            return {k: np.asarray(z[k]) for k in z.files}
    return tf_bundle.read_bundle(path)


def average(checkpoints):
    """[checkpoint prefix] -> {name: averaged array} (global_step* skipped), reference arithmetic."""
    if not checkpoints:
        raise ValueError('No checkpoints provided for averaging.')
    first = _read_all(checkpoints[0])
    acc = {n: np.zeros(a.shape) for n, a in first.items() if not n.startswith('global_step')}
    dtypes = {}
    for i, c in enumerate(checkpoints):
        cur = first if i == 0 else _read_all(c)
        for n in acc:
            if n not in cur:
                raise KeyError('%s is missing from %s' % (n, c))
            dtypes[n] = cur[n].dtype
            acc[n] += cur[n]
    return {n: (v / len(checkpoints)).astype(dtypes[n] if dtypes[n].kind == 'f' else np.float32) for n, v in acc.items()}


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split('\n')[0])
    ap.add_argument('--checkpoints', required=True, help='Comma-separated list of checkpoints to average.')
    ap.add_argument('--prefix', default='', help='Prefix (e.g., directory) to append to each checkpoint.')
    ap.add_argument('--output_path', default='/tmp/averaged.ckpt')
    ap.add_argument('--format', default='both', choices=['tf', 'npz', 'both'])
    args = ap.parse_args(argv)
    ckpts = [c.strip() for c in args.checkpoints.split(',') if c.strip()]
    if not ckpts:
        raise ValueError('No checkpoints provided for averaging.')
    ckpts = [args.prefix + c for c in ckpts]
    ckpts = [c for c in ckpts if checkpoint_exists(c)]
    if not ckpts:
        raise ValueError('None of the provided checkpoints exist. %s' % args.checkpoints)
    avg = average(ckpts)
    avg['global_step'] = np.zeros((), np.int64)
    out = args.output_path + '-0'
    if args.format in ('tf', 'both'):
        tf_bundle.write_bundle(out, avg)
    if args.format in ('npz', 'both'):
        np.savez(out + '.npz', **avg)
    tf_bundle.write_checkpoint_state(os.path.dirname(out) or '.', os.path.basename(out))
    print('Averaged checkpoints saved in %s' % out)
    return out


if __name__ == '__main__':
    main()
