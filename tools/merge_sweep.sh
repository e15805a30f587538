#!/bin/bash
for m in ${MERGES:-0 32 64 96}; do
  timeout -k 10 300 python bench.py --workload bucketed --vocab 1388 --merge-utts $m --no-cpu-baseline --no-alt-precisions > gpurun_out/mg_tmp.log 2>&1 || { tail -5 gpurun_out/mg_tmp.log; exit 1; }
  python - $m <<'PY'
import json,sys
for l in open('gpurun_out/mg_tmp.log'):
    if l.startswith('{'):
        d=json.loads(l); print('merge', sys.argv[1], round(d['value'],1), round(d['ms_per_step'],1), d['config'].get('joint_decodes'), d['hypotheses_crc32'], flush=True)
PY
done
