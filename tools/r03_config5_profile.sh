#!/bin/bash
# development helper (GPU box): kernel stats of BASELINE config 5 (50 x 30 s, V = 3205) in a given precision
# usage: tools/r03_config5_profile.sh PRECISION OUT.csv
set -u
R=$PWD; prec=${1:-bf16}; out=${2:-gpurun_out/r03/config5_$prec.csv}
mkdir -p $(dirname $out)
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c5_$prec -o ks -- python3 $R/bench.py --batch 50 --t-raw 3000 --vocab 3205 --precision $prec --steps 8 --warmup 1 --no-cpu-baseline --no-alt-precisions --no-one-at-a-time --no-configs > $R/${out%.csv}.json 2> /tmp/c5.err || { tail -5 /tmp/c5.err; exit 1; }
cp $(find /tmp/c5_$prec -name 'ks_kernel_stats.csv' | head -1) $R/$out
