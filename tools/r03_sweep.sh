#!/bin/bash
# development helper (GPU box): bench.py headline under a list of settings "ENV=.. ENV=.. -- bench args", one line per setting
# usage: tools/r03_sweep.sh 'GPU_MAX_HW_QUEUES=16 -- --in-flight 6' 'ASR_AMD_SPLITS=1,1,4 --' ...
for spec in "$@"; do
  envs="${spec%%--*}"; args="${spec#*--}"
  env $envs timeout -k 10 200 python bench.py --steps 8 --warmup 1 --no-cpu-baseline --no-alt-precisions --no-one-at-a-time --no-configs $args > gpurun_out/sweep_tmp.log 2>&1 || { echo "FAILED: $spec"; tail -3 gpurun_out/sweep_tmp.log; continue; }
  python - "$spec" <<'PY'
import json,sys
for l in open('gpurun_out/sweep_tmp.log'):
    if l.startswith('{'):
        d=json.loads(l); print('%-60s %8.1f audio-s/s %8.2f ms/step  hwq %s' % (sys.argv[1], d['value'], d['ms_per_step'], d['config'].get('hw_queues')), flush=True)
PY
done
