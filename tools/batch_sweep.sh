#!/bin/bash
run() { timeout -k 10 300 python bench.py --no-cpu-baseline --no-alt-precisions --no-one-at-a-time "$@" > gpurun_out/bb_tmp.log 2>&1 || { tail -5 gpurun_out/bb_tmp.log; exit 1; }
  python - "$*" <<'PY'
import json,sys
for l in open('gpurun_out/bb_tmp.log'):
    if l.startswith('{'):
        d=json.loads(l); print(sys.argv[1], '->', round(d['value'],1), round(d['ms_per_step'],2), flush=True)
PY
}
run --batch 64 --in-flight 4 --steps 8 && run --batch 128 --in-flight 2 --engines-per-gpu 2 --steps 4 && run --batch 128 --in-flight 4 --steps 8 && run --batch 256 --in-flight 1 --engines-per-gpu 1 --steps 2 && run --batch 256 --in-flight 2 --engines-per-gpu 2 --steps 4 && run --batch 128 --in-flight 3 --engines-per-gpu 3 --steps 6 \
 && run --batch 96 --in-flight 4 --steps 8 && run --batch 192 --in-flight 4 --steps 8 && run --batch 256 --in-flight 4 --steps 4 && run --batch 256 --in-flight 3 --engines-per-gpu 3 --steps 3
