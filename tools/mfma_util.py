#!/usr/bin/env python3
"""Summarise one rocprofv3 --pmc pass of SQ / GRBM counters per kernel class -> profiles/rNN_mfma_util.json.

On the GPU box (a pass of its own, --kernel-trace only, the program directly after `--`; 8 SQ + 1 GRBM counters fit one
pass on gfx950):

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 \
        SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE \
        --kernel-trace --output-format csv -d DIR -o mfma -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-alt-precisions
    python3 tools/mfma_util.py DIR/mfma_counter_collection.csv profiles/r02_mfma_util.json

(the raw CSV is > 100 MB: summarise it on the box and bring back only the JSON).

Derived per class (sums over its launches):
  mfma_util        = SQ_VALU_MFMA_BUSY_CYCLES / ((GRBM_GUI_ACTIVE / 8) * 1024)   rocprofv3's own MfmaUtil expression: busy cycles summed
                     over the 1024 SIMDs, GRBM_GUI_ACTIVE summed over the 8 XCDs (MI355X_MICROARCH.md, DVFS give-back)
  mfma_tflops      = 512 * (MOPS_F16 + MOPS_F32) / wall time            executed matrix FLOP/s (MOPS counts add-or-mul ops / 512)
  eff_clock_ghz    = (GRBM_GUI_ACTIVE / 8) / wall time                  the clock the chip held (reads high on short dispatches)
  valu_per_mfma_cycle = 4 * SQ_ACTIVE_INST_VALU / SQ_VALU_MFMA_BUSY_CYCLES
"""
import collections
import csv
import json
import sys

sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.abspath(__file__)))
from pmc_traffic import bench_class  # noqa: E402


def main(path, out_path):
    cls = collections.defaultdict(lambda: collections.defaultdict(float))
    seen = set()
    with open(path, newline='') as f:
        for r in csv.DictReader(f):
            c = cls[bench_class(r['Kernel_Name'])]
            c[r['Counter_Name']] += float(r['Counter_Value'])
            did = r['Dispatch_Id']
            if did not in seen:
                seen.add(did)
                c['_launches'] += 1
                c['_ns'] += float(r['End_Timestamp']) - float(r['Start_Timestamp'])
    res = {'_about': 'rocprofv3 --pmc (SQ_* + GRBM_GUI_ACTIVE, one pass, --kernel-trace only) over `python3 bench.py --steps 1 --warmup 0 '
                     '--no-cpu-baseline --no-alt-precisions` (two decode passes of config 2, default precision f32_f16x3, two half '
                     'batches of 320 beam rows).  Sums over the launches of a class; see tools/mfma_util.py for the derived fields. '
                     'Counter collection serialises the dispatches, so the halves do not overlap in this run.'}
    order = sorted(cls.items(), key=lambda kv: -kv[1]['_ns'])
    for name, c in order:
        n, ns = c['_launches'], c['_ns']
        gui = c.get('GRBM_GUI_ACTIVE', 0.0) / 8.0
        busy = c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0)
        mops = c.get('SQ_INSTS_VALU_MFMA_MOPS_F16', 0.0) + c.get('SQ_INSTS_VALU_MFMA_MOPS_F32', 0.0)
        e = {'launches': int(n), 'avg_us': round(ns / n / 1e3, 2), 'total_ms': round(ns / 1e6, 2),
             'mfma_util': round(busy / (gui * 1024.0), 4) if gui else None,
             'mfma_tflops': round(512.0 * mops / ns / 1e3, 1) if ns else None,
             'eff_clock_ghz': round(gui / ns, 2) if ns else None,
             'valu_per_mfma_cycle': round(4.0 * c.get('SQ_ACTIVE_INST_VALU', 0.0) / busy, 2) if busy else None,
             'mfma_coexec_frac': round(c.get('SQ_VALU_MFMA_COEXEC_CYCLES', 0.0) / busy, 3) if busy else None,
             'counters': {k: int(v) for k, v in sorted(c.items()) if not k.startswith('_')}}
        res[name] = e
    json.dump(res, open(out_path, 'w'), indent=1)
    print('wrote', out_path, len(res) - 1, 'kernel classes')
    for name, c in order[:10]:
        e = res[name]
        print('%-34s n=%6d avg %7.2f us  mfma_util %s  mfma TF/s %s  clk %s' % (name, e['launches'], e['avg_us'], e['mfma_util'], e['mfma_tflops'], e['eff_clock_ghz']))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
