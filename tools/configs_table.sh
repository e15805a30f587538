#!/bin/bash
# development helper (GPU box): the BASELINE configurations of DESIGN.md section 7, one bench line each
run() { name=$1; shift
  timeout -k 10 400 python bench.py --no-cpu-baseline --no-alt-precisions "$@" > gpurun_out/cfg_$name.log 2>&1 || { tail -5 gpurun_out/cfg_$name.log; exit 1; }
  python - $name <<'PY'
import json,sys
for l in open('gpurun_out/cfg_%s.log' % sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); o=d.get('one_step_at_a_time') or {}
        print(sys.argv[1], round(d['value'],1), round(d['ms_per_step'],2), 'one at a time', round(o.get('value',0),1), flush=True)
PY
}
run word --vocab 28444 && run c3 --workload bucketed --vocab 1388 && run c4 --workload bucketed --vocab 28444 && run c5 --batch 50 --t-raw 3000 --vocab 3205 --precision bf16 && run c5f32 --batch 50 --t-raw 3000 --vocab 3205
