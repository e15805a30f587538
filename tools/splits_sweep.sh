#!/bin/bash
# bf16 mode: K-slice policy sweep (qkv,proj,ffn2), four steps in flight
for sp in 1,1,2 1,2,4 2,2,4 2,2,8 1,1,1 1,1,4; do
  ASR_AMD_SPLITS=$sp timeout -k 10 150 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-alt-precisions --no-one-at-a-time --precision bf16 > gpurun_out/sp_tmp.log 2>&1 || { tail -5 gpurun_out/sp_tmp.log; exit 1; }
  python - $sp <<'PY'
import json,sys
for l in open('gpurun_out/sp_tmp.log'):
    if l.startswith('{'):
        d=json.loads(l); print(sys.argv[1], round(d['value'],1), round(d['ms_per_step'],2), flush=True)
PY
done
