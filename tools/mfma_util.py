#!/usr/bin/env python3
"""Summarise one rocprofv3 --pmc pass of SQ / GRBM counters per kernel class -> profiles/rNN_mfma_util.json.

On the GPU box (a pass of its own, --kernel-trace only, the program directly after `--`; 8 SQ + 1 GRBM counters fit one
pass on gfx950):

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 \
        SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE \
        --kernel-trace --output-format csv -d DIR -o mfma -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-alt-precisions
    python3 tools/mfma_util.py DIR/mfma_counter_collection.csv profiles/r03_mfma_util.json 'the profiled command'

(the raw CSV is > 100 MB: summarise it on the box and bring back only the JSON).

Derived per class and per (class, workgroups of the launch) - decode-row and encoder launches share templates:
  mfma_busy_of_peak = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x dispatch time x 2.4 GHz): the share of the chip's matrix-pipe
                     cycles AT ITS PEAK CLOCK that the launches kept busy, from the dispatch's own begin / end timestamps.
                     This is the figure to quote.  (Round 2 divided by GRBM_GUI_ACTIVE / 8 instead, rocprofv3's own MfmaUtil
                     expression; that counter keeps running while a dispatch drains and gave "clocks" of 2.7-5.6 GHz on
                     dispatches of a few microseconds, so it under-stated short kernels.  It is kept as
                     `mfma_util_by_gui_active` for comparison and no clock is derived from it any more.)
  mfma_tflops      = 512 * (MOPS_F16 + MOPS_F32) / dispatch time        executed matrix FLOP/s (MOPS counts add-or-mul ops / 512)
  valu_per_mfma_cycle = 4 * SQ_ACTIVE_INST_VALU / SQ_VALU_MFMA_BUSY_CYCLES
"""
import collections
import csv
import json
import sys

sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.abspath(__file__)))
from pmc_traffic import bench_class, shape_of, workgroups  # noqa: E402

PEAK_CLOCK_GHZ = 2.4


def derive(c):
    n, ns = c['_launches'], c['_ns']
    gui = c.get('GRBM_GUI_ACTIVE', 0.0) / 8.0
    busy = c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0)
    mops = c.get('SQ_INSTS_VALU_MFMA_MOPS_F16', 0.0) + c.get('SQ_INSTS_VALU_MFMA_MOPS_F32', 0.0)
    return {'launches': int(n), 'avg_us': round(ns / n / 1e3, 2), 'total_ms': round(ns / 1e6, 2),
            'mfma_busy_of_peak': round(busy / (1024.0 * ns * PEAK_CLOCK_GHZ), 4) if ns else None,
            'mfma_util_by_gui_active': round(busy / (gui * 1024.0), 4) if gui else None,
            'mfma_tflops': round(512.0 * mops / ns / 1e3, 1) if ns else None,
            'valu_per_mfma_cycle': round(4.0 * c.get('SQ_ACTIVE_INST_VALU', 0.0) / busy, 2) if busy else None,
            'mfma_coexec_frac': round(c.get('SQ_VALU_MFMA_COEXEC_CYCLES', 0.0) / busy, 3) if busy else None}


def main(path, out_path):
    cls = collections.defaultdict(lambda: collections.defaultdict(float))
    grids = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
    seen = set()
    with open(path, newline='') as f:
        for r in csv.DictReader(f):
            name = bench_class(r['Kernel_Name'])
            for c in (cls[name], grids[name][workgroups(r)]):
                c[r['Counter_Name']] += float(r['Counter_Value'])
            did = r['Dispatch_Id']
            if did not in seen:
                seen.add(did)
                for c in (cls[name], grids[name][workgroups(r)]):
                    c['_launches'] += 1
                    c['_ns'] += float(r['End_Timestamp']) - float(r['Start_Timestamp'])
    about = sys.argv[3] if len(sys.argv) > 3 else '(command not recorded)'
    res = {'_about': 'rocprofv3 --pmc (SQ_* + GRBM_GUI_ACTIVE, one pass, --kernel-trace only) over `%s`.  Sums over the launches of '
                     'a class; `by_grid` keeps launches of one template apart by workgroup count (decode rows vs encoder).  See '
                     'tools/mfma_util.py for the derived fields.  Counter collection serialises the dispatches: every kernel runs '
                     'alone in this run.' % about}
    order = sorted(cls.items(), key=lambda kv: -kv[1]['_ns'])
    for name, c in order:
        e = derive(c)
        e['counters'] = {k: int(v) for k, v in sorted(c.items()) if not k.startswith('_')}
        e['by_grid'] = {('%d workgroups' % w) + ((' = ' + shape_of(name, w)) if shape_of(name, w) else ''): derive(g)
                        for w, g in sorted(grids[name].items(), key=lambda kv: -kv[1]['_ns'])}
        res[name] = e
    json.dump(res, open(out_path, 'w'), indent=1)
    print('wrote', out_path, len(res) - 1, 'kernel classes')
    for name, c in order[:10]:
        e = res[name]
        print('%-34s n=%6d avg %7.2f us  mfma busy of peak %s  mfma TF/s %s' % (name, e['launches'], e['avg_us'], e['mfma_busy_of_peak'], e['mfma_tflops']))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
