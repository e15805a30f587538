#!/bin/bash
# Round-3 evidence (run on the GPU box from the repo root): kernel stats, kernel trace overlap, PMC traffic (two precisions),
# matrix-pipe counters.  Every rocprofv3 run is a run of its own, the program directly after `--`, counters only with
# --kernel-trace.  Summaries land in gpurun_out/r03/prof/ (copy what is to be judged into profiles/).
set -u
R=$PWD
O=$R/gpurun_out/r03/prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py --steps 8 --warmup 1 --no-cpu-baseline --no-alt-precisions --no-one-at-a-time --no-configs"
CMD="python3 bench.py --steps 8 --warmup 1 --no-cpu-baseline --no-alt-precisions --no-one-at-a-time --no-configs"
step() { echo "== $1" >&2; }

step "kernel stats"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r03_ks -o ks -- python3 $B > $O/bench_line_under_rocprof.json 2> /tmp/r03_ks.err || { tail -5 /tmp/r03_ks.err; exit 1; }
f=$(find /tmp/r03_ks -name 'ks_kernel_stats.csv' | head -1); cp "$f" $O/r03_bench_kernel_stats_v19.csv
t=$(find /tmp/r03_ks -name 'ks_kernel_trace.csv' | head -1)
python3 $R/tools/trace_overlap.py "$t" $O/r03_trace_overlap.json > /dev/null || exit 1

for prec in f32_f16x3 bf16; do
  for c in FETCH_SIZE WRITE_SIZE; do
    n=$( [ $c = FETCH_SIZE ] && echo fetch || echo write )
    step "pmc $c $prec"
    timeout -s KILL 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/r03_pmc_$prec -o $n -- python3 $B --precision $prec > /dev/null 2> /tmp/r03_pmc.err || { tail -5 /tmp/r03_pmc.err; exit 1; }
  done
  d=$(dirname $(find /tmp/r03_pmc_$prec -name 'fetch_counter_collection.csv' | head -1))
  w=$(dirname $(find /tmp/r03_pmc_$prec -name 'write_counter_collection.csv' | head -1))
  [ "$d" != "$w" ] && cp $w/write_counter_collection.csv $d/
  python3 $R/tools/pmc_traffic.py $d $O/r03_${prec}_pmc_traffic.json "$CMD --precision $prec" || exit 1
done

step "mfma counters"
timeout -s KILL 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/r03_mfma -o mfma -- python3 $B > /dev/null 2> /tmp/r03_mfma.err || { tail -5 /tmp/r03_mfma.err; exit 1; }
python3 $R/tools/mfma_util.py $(find /tmp/r03_mfma -name 'mfma_counter_collection.csv' | head -1) $O/r03_mfma_util.json "$CMD" > $O/r03_mfma_util.txt || exit 1
ls -la $O >&2
