#!/usr/bin/env python3
"""Summarise rocprofv3 PMC passes into profiles/rNN_pmc_traffic.json (what bench.py reports as roofline.traffic).

On the GPU box (separate passes, as /opt/skills/guides/MI355X_MICROARCH.md prescribes; --kernel-trace only):

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/prof -o fetch -- \
        python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/prof -o write -- \
        python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline

then here:  python tools/pmc_traffic.py gpurun_out/prof profiles/r01_pmc_traffic.json
"""
import collections
import csv
import json
import re
import sys

EPI = {'0': 'BIAS', '1': 'GLU', '2': 'PARTIAL'}
RENAME = {'add_layer_norm_row': 'add_layer_norm(dec)', 'input_norm_apply': 'input_norm_timing'}


def per_kernel(path):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        a = acc[r['Kernel_Name']]
        a[0] += 1
        a[1] += float(r['Counter_Value'])
    return acc


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
    for k, (n, kb) in fetch.items():
        e = cls.setdefault(bench_class(k), {'launches': 0, 'fetch_kb': 0.0, 'write_kb': 0.0})
        e['launches'] += n
        e['fetch_kb'] += kb
        e['write_kb'] += write.get(k, [0, 0.0])[1]
    res = {'_about': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) over '
                     '`python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline` (two decode passes of config 2). '
                     'Per-launch averages, keyed by bench.py kernel class. traffic_bytes = 2 x FETCH_SIZE + WRITE_SIZE '
                     '(KB -> bytes): on gfx950 FETCH_SIZE reports half of the bytes of 16-B-per-lane streaming reads and '
                     'WRITE_SIZE is exact (MI355X_MICROARCH.md, HBM section); Infinity-Cache hits count as traffic.'}
    for c, e in sorted(cls.items(), key=lambda kv: -kv[1]['fetch_kb']):
        n = e['launches']
        res[c] = {'launches': n, 'FETCH_SIZE_KB_per_launch': round(e['fetch_kb'] / n, 1),
                  'WRITE_SIZE_KB_per_launch': round(e['write_kb'] / n, 1),
                  'traffic_bytes_per_launch': int((2 * e['fetch_kb'] + e['write_kb']) / n * 1024)}
    json.dump(res, open(out_path, 'w'), indent=1)
    print('wrote', out_path, len(res) - 1, 'kernel classes')


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
