#!/bin/bash
# Dev helper for gpurun: run the given commands one after another (each "LIMIT_SECONDS::command"), log to gpurun_out/,
# and stop at the first one that was killed at its limit (a hung GPU step must not be followed by another).
mkdir -p gpurun_out
i=0
for spec in "$@"; do
    lim="${spec%%::*}"; cmd="${spec#*::}"
    i=$((i+1))
    echo "[gpu_seq] step $i (limit ${lim}s): $cmd"
    timeout -k 10 "$lim" bash -c "$cmd"
    rc=$?
    echo "[gpu_seq] step $i rc=$rc"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "[gpu_seq] step $i killed at its limit: stopping"; exit $rc; fi
done
exit 0
