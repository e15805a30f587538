#!/bin/bash
for n in 4 6 8; do
  timeout -k 10 300 python bench.py --workload bucketed --vocab 1388 --engines-per-gpu $n --no-cpu-baseline --no-alt-precisions > gpurun_out/bk_tmp.log 2>&1 || { tail -5 gpurun_out/bk_tmp.log; exit 1; }
  python - $n <<'PY'
import json,sys
for l in open('gpurun_out/bk_tmp.log'):
    if l.startswith('{'):
        d=json.loads(l); print('engines', sys.argv[1], round(d['value'],1), round(d['ms_per_step'],1), {k:d['config'].get(k) for k in ('batches','engines_per_gpu')}, flush=True)
PY
done
