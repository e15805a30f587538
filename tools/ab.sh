#!/bin/bash
# development helper (run on the GPU box): A/B the bench between ab/libasr_prev.so and the in-tree build, interleaved
# usage: tools/ab.sh [rounds] [extra bench args]
n=${1:-2}; shift
for i in $(seq $n); do
  for lib in ab/libasr_prev.so asr_amd/csrc/libasr_amd.so; do
    ASR_AMD_LIB=$PWD/$lib timeout -k 10 150 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-alt-precisions "$@" > gpurun_out/ab_tmp.log 2>&1 || { tail -5 gpurun_out/ab_tmp.log; exit 1; }
    python - $lib <<'PY'
import json,sys
for l in open('gpurun_out/ab_tmp.log'):
    if l.startswith('{'):
        d=json.loads(l); print(sys.argv[1], round(d['value'],1), round(d['ms_per_step'],2), 'kernel-sum', round(sum(k['ms'] for k in d['kernels']),1), flush=True)
PY
  done
done
