for sp in "0,0,0" "2,4,4" "2,4,8" "4,4,8" "0,0,0"; do
  ASR_AMD_SPLITS=$sp timeout -k 10 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/sp.log 2>&1 || exit 1
  python - "$sp" <<'PY'
import json,sys
for l in open('gpurun_out/sp.log'):
    if l.startswith('{'):
        d=json.loads(l); print(sys.argv[1], round(d['value'],1), flush=True)
PY
done
