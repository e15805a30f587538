#!/bin/bash
# development helper (GPU box): the two PMC traffic passes of tools/r03_profiles.sh for ONE precision
# usage: tools/r03_pmc_one.sh PRECISION
set -u
R=$PWD; prec=${1:-bf16}; O=$R/gpurun_out/r03/prof; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py --steps 8 --warmup 1 --no-cpu-baseline --no-alt-precisions --no-one-at-a-time --no-configs"
CMD="python3 bench.py --steps 8 --warmup 1 --no-cpu-baseline --no-alt-precisions --no-one-at-a-time --no-configs"
rm -rf /tmp/r03_pmc_$prec
for c in FETCH_SIZE WRITE_SIZE; do
  n=$( [ $c = FETCH_SIZE ] && echo fetch || echo write )
  timeout -s KILL 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/r03_pmc_$prec -o $n -- python3 $B --precision $prec > /dev/null 2> /tmp/r03_pmc.err || { tail -5 /tmp/r03_pmc.err; exit 1; }
done
d=$(dirname $(find /tmp/r03_pmc_$prec -name 'fetch_counter_collection.csv' | head -1))
w=$(dirname $(find /tmp/r03_pmc_$prec -name 'write_counter_collection.csv' | head -1))
[ "$d" != "$w" ] && cp $w/write_counter_collection.csv $d/
python3 $R/tools/pmc_traffic.py $d $O/r03_${prec}_pmc_traffic.json "$CMD --precision $prec"
