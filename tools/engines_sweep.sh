#!/bin/bash
# development helper (GPU box): steps in flight sweep; usage: tools/engines_sweep.sh "<n list>" [bench args]
ns=$1; shift
for n in $ns; do
  timeout -k 10 200 python bench.py --steps $((2*n)) --warmup 1 --no-cpu-baseline --no-alt-precisions --no-one-at-a-time --engines-per-gpu $n --in-flight $n "$@" > gpurun_out/en_tmp.log 2>&1 || { tail -5 gpurun_out/en_tmp.log; exit 1; }
  python - $n <<'PY'
import json,sys
for l in open('gpurun_out/en_tmp.log'):
    if l.startswith('{'):
        d=json.loads(l); print('engines', sys.argv[1], round(d['value'],1), round(d['ms_per_step'],2), flush=True)
PY
done
