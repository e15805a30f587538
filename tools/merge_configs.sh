#!/bin/bash
run() { name=$1; shift
  timeout -k 10 400 python bench.py --no-cpu-baseline --no-alt-precisions "$@" > gpurun_out/cfgm_$name.log 2>&1 || { tail -5 gpurun_out/cfgm_$name.log; exit 1; }
  python - $name <<'PY'
import json,sys
for l in open('gpurun_out/cfgm_%s.log' % sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l)
        print(sys.argv[1], round(d['value'],1), round(d['ms_per_step'],2), d['config'].get('joint_decodes'), d.get('hypotheses_crc32'), flush=True)
PY
}
run c1 --workload bucketed --utts-per-gpu 8 --beam 1 && run c1m --workload bucketed --utts-per-gpu 8 --beam 1 --merge-utts 8 \
 && run c4 --workload bucketed --vocab 28444 && run c4m64 --workload bucketed --vocab 28444 --merge-utts 64 && run c4m192 --workload bucketed --vocab 28444 --merge-utts 192 \
 && run c3bf --workload bucketed --vocab 1388 --precision bf16 && run c3bfm --workload bucketed --vocab 1388 --precision bf16 --merge-utts 192
