#!/bin/bash
# development helper (GPU box): sample the GPU's clocks and power while a command runs
# usage: tools/power_probe.sh OUT_FILE command...
out=$1; shift
( while true; do rocm-smi --showpower --showclocks --showuse --json 2>/dev/null | tr -d '\n'; echo; sleep 0.2; done > "$out" ) &
sampler=$!
"$@"
rc=$?
kill $sampler 2>/dev/null
exit $rc
