#!/usr/bin/env python3
"""Summarise rocprofv3 PMC passes into profiles/rNN_pmc_traffic.json (what bench.py reports as roofline.traffic).

On the GPU box (separate passes, as /opt/skills/guides/MI355X_MICROARCH.md prescribes; --kernel-trace only):

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/prof -o fetch -- \
        python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/prof -o write -- \
        python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline

then here:  python tools/pmc_traffic.py gpurun_out/prof profiles/r03_f32_f16x3_pmc_traffic.json 'the profiled command'
"""
import collections
import csv
import json
import re
import sys

EPI = {'0': 'BIAS', '1': 'GLU', '2': 'PARTIAL'}
RENAME = {'add_layer_norm_row': 'add_layer_norm(dec)', 'input_norm_apply': 'input_norm_timing'}


def workgroups(r):
    """Workgroups of the dispatch of CSV row r (Grid_Size counts work-items)."""
    try:
        return int(r['Grid_Size']) // max(1, int(r['Workgroup_Size']))
    except (KeyError, ValueError):
        return 0


def per_kernel(path):
    """(kernel name, workgroups of the launch) -> [launches, summed counter]: a template is launched on several shapes (the
    decode rows' 640 x N x K and the encoder's 21376 x N x K share `gemm_split_nt_kernel<2,1,8,...>`), and their traffic per
    launch differs by a factor of 30, so launches are kept apart by their grid."""
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        a = acc[(r['Kernel_Name'], workgroups(r))]
        a[0] += 1
        a[1] += float(r['Counter_Value'])
    return acc


# GEMM shapes of the char model at config 2 as bench.py's default run launches them (two 64-utterance steps per call: 1280
# decode rows; 21376 encoder rows per step, 42752 per call; M x N x K, N of the GLU layer counts value and gate columns),
# for naming a grid.  PMC_ROWS=640 in the environment names the grids of a `--steps-per-decode 1` run instead.
import os as _os
_R = int(_os.environ.get('PMC_ROWS', '1280'))
_E = 21376 * (_R // 640)
MODEL_SHAPES = [(_R, 3072, 1024), (_R, 1024, 1024), (_R, 8192, 1024), (_R, 1024, 4096), (_R, 3900, 1024),
                (_E, 1024, 320), (_E, 3072, 1024), (_E, 1024, 1024), (_E, 8192, 1024), (_E, 1024, 4096), (_E, 2048, 1024)]


def shape_of(cls, wgs):
    """Best-effort (M, N, K, K-slices) of a GEMM launch of class `cls` = 'name<BM,BN,EPI>' with `wgs` workgroups: the model
    shapes whose tile count x K-slices equals the grid; None when that is not unique among shapes of the same epilogue."""
    m = re.match(r'gemm_\w+<(\d+),(\d+),(\w+)>', cls)
    if not m or not wgs:
        return None
    bm, bn, epi = int(m.group(1)), int(m.group(2)), m.group(3)
    hits = []
    for (M, N, K) in MODEL_SHAPES:
        if (epi == 'GLU') != (N == 8192):
            continue
        nt_m, nt_n = -(-M // bm), -(-N // bn)
        for split in ((1,) if epi != 'PARTIAL' else (1, 2, 4, 8)):
            grids = {nt_m * nt_n * split}                      # the legacy order: one workgroup per tile and K-slice
            if nt_m <= 32:                                     # the per-XCD partition pads every XCD to the largest share
                for pz in (1, 2, 4, 8):
                    for pm in (1, 2, 4, 8):
                        if split % pz == 0 and pm * pz <= 8:
                            pn = 8 // (pm * pz)
                            if pm <= nt_m and pn <= nt_n:
                                grids.add(8 * -(-nt_m // pm) * -(-nt_n // pn) * (split // pz))
            if wgs in grids:
                hits.append('%dx%dx%d' % (M, N, K) + ('/%d' % split if split > 1 else ''))
    return hits[0] if len(hits) == 1 else ('|'.join(hits) if hits else None)


def bench_class(kernel):
    m = re.search(r'gemm_f32_nt_kernel<(\d+), (\d+), \d+, \d+, (\d)(?:, (\w+))?>', kernel)
    if m:
        return 'gemm_f32_nt<%s,%s,%s>' % (m.group(1), m.group(2), EPI.get(m.group(3), 'ACT'))
    m = re.search(r'gemm_split_nt_kernel<(\d+), (\d+), (\d), (\d), (\d+)>', kernel)   # <MI, NI, NW, EPI, NPROD>: 32 MI x 32 NI NW tile
    if m:
        base = {'13': 'gemm_f16x3_nt', '6': 'gemm_bf16x6_nt', '3': 'gemm_bf16x3_nt', '1': 'gemm_bf16_nt'}[m.group(5)]
        return '%s<%d,%d,%s>' % (base, 32 * int(m.group(1)), 32 * int(m.group(2)) * int(m.group(3)), EPI.get(m.group(4), 'ACT'))
    m = re.search(r'::(\w+?)(?:_kernel)?(?:<[^>]*>)?\(', kernel) or re.search(r'(\w+)', kernel)
    return RENAME.get(m.group(1), m.group(1))


def main(prof_dir, out_path):
    fetch = per_kernel(prof_dir + '/fetch_counter_collection.csv')
    write = per_kernel(prof_dir + '/write_counter_collection.csv')
    cls = {}
    for (k, wgs), (n, kb) in fetch.items():
        c = bench_class(k)
        e = cls.setdefault(c, {'launches': 0, 'fetch_kb': 0.0, 'write_kb': 0.0, 'grids': {}})
        wkb = write.get((k, wgs), [0, 0.0])[1]
        e['launches'] += n
        e['fetch_kb'] += kb
        e['write_kb'] += wkb
        g = e['grids'].setdefault(wgs, {'launches': 0, 'fetch_kb': 0.0, 'write_kb': 0.0})
        g['launches'] += n
        g['fetch_kb'] += kb
        g['write_kb'] += wkb
    about = sys.argv[3] if len(sys.argv) > 3 else '(command not recorded)'
    res = {'_about': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) over `%s`. '
                     'Per-launch averages, keyed by bench.py kernel class; `by_grid` keeps the launches of one template apart by '
                     'their workgroup count (decode-row and encoder launches share templates) and names the GEMM shape '
                     'M x N x K[/K-slices] where the grid identifies it. traffic_bytes = 2 x FETCH_SIZE + WRITE_SIZE '
                     '(KB -> bytes): on gfx950 FETCH_SIZE reports half of the bytes of 16-B-per-lane streaming reads and '
                     'WRITE_SIZE is exact (MI355X_MICROARCH.md, HBM section); Infinity-Cache hits count as traffic.' % about}
    for c, e in sorted(cls.items(), key=lambda kv: -kv[1]['fetch_kb']):
        n = e['launches']
        res[c] = {'launches': n, 'FETCH_SIZE_KB_per_launch': round(e['fetch_kb'] / n, 1),
                  'WRITE_SIZE_KB_per_launch': round(e['write_kb'] / n, 1),
                  'traffic_bytes_per_launch': int((2 * e['fetch_kb'] + e['write_kb']) / n * 1024),
                  'by_grid': {('%d workgroups' % w) + ((' = ' + shape_of(c, w)) if shape_of(c, w) else ''):
                              {'launches': g['launches'],
                               'traffic_bytes_per_launch': int((2 * g['fetch_kb'] + g['write_kb']) / g['launches'] * 1024)}
                              for w, g in sorted(e['grids'].items(), key=lambda kv: -kv[1]['launches'])}}
    json.dump(res, open(out_path, 'w'), indent=1)
    print('wrote', out_path, len(res) - 1, 'kernel classes')


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
