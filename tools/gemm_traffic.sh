#!/bin/bash
# development helper (GPU box): FETCH_SIZE / WRITE_SIZE per launch and the duration of the decode-row GEMM shapes in
# tools/gemm_split_bench (cold weights), with the per-XCD tile partition (default) and without (ASR_AMD_XCD_MAP=0)
# usage: tools/gemm_traffic.sh "dglu dqkv dproj dffn2" [maps, default "0 1"]
R=$PWD; cd /tmp; export TMPDIR=/tmp
for m in ${2:-0 1}; do for sh in $1; do
  t=9; case $sh in proj) t=2;; esac
  sp=1; case $sh in ffn2|dffn2) sp=2;; esac
  for c in FETCH_SIZE WRITE_SIZE; do
    ASR_AMD_XCD_MAP=$m timeout -s KILL 60 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/gt_${m}_${sh}_$c -o x -- $R/tools/gemm_split_bench 20 $sh 13 $t $sp 1 > /tmp/gt.out 2>&1 || echo fail $m $sh
  done
  us=$(grep "tile=$t split= *$sp " /tmp/gt.out | awk '{for(i=1;i<=NF;i++) if($i=="us") print $(i-1)}' | head -1)
  python3 - $m $sh "$us" <<'PY'
import csv,glob,sys
m,sh,us=sys.argv[1:4]
tot={}
for c in ('FETCH_SIZE','WRITE_SIZE'):
    f=glob.glob('/tmp/gt_%s_%s_%s/**/x_counter_collection.csv'%(m,sh,c),recursive=True)
    n=0;v=0.0
    for r in csv.DictReader(open(f[0])):
        if 'gemm_split' in r['Kernel_Name']: n+=1; v+=float(r['Counter_Value'])
    tot[c]=v/max(n,1)*1024/1e6
print('map %s %-6s fetch x2 %7.1f MB  write %6.1f MB  traffic %7.1f MB per launch   %s us (under the counter run)' % (m,sh,2*tot['FETCH_SIZE'],tot['WRITE_SIZE'],2*tot['FETCH_SIZE']+tot['WRITE_SIZE'],us))
PY
done; done
