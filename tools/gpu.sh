#!/bin/bash
# development helper: run a command on the GPU pool, waiting (bounded) while no box is free (exit 3 = nothing ran)
# usage: tools/gpu.sh TIMEOUT_SECONDS 'command'
t=$1; shift
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do
    /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"
    rc=$?
    [ $rc -ne 3 ] && exit $rc
    sleep 45
done
exit 3
