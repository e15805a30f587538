#!/bin/bash
# What each decode kernel class costs with everything else as it is: bench.py with that class launched twice
# (ASR_AMD_DOUBLE, include/asr_amd.h has no part in it: a measurement aid of the engine).  Usage (GPU box):
#   tools/marginal_cost.sh [in-flight, default 4] > gpurun_out/marginal_cost.txt
IF=${1:-4}
X="--no-cpu-baseline --no-alt-precisions --no-one-at-a-time --no-configs --steps 16 --in-flight $IF"
for c in none glu proj qkv ffn2 logits self cross ln topk encoder none; do
    ASR_AMD_DOUBLE=$c timeout -k 10 250 python bench.py $X > /tmp/mc.json 2>/dev/null || exit 1
    python -c "
import json
d=json.load(open('/tmp/mc.json')); print('in-flight $IF, twice: %-7s %.1f ms per 64-utterance step' % ('$c', d['ms_per_step']))"
done
