#!/bin/bash
# Run one GPU step under its own time limit and log file; a step that is KILLED at its limit (rc 124 / 137) stops the chain
# (a hung GPU step must not be followed by another one in the same gpurun call), an ordinary failure does not.
#   scripts/gpu_step.sh <log file> <seconds> <command...>
log=$1; lim=$2; shift 2
mkdir -p "$(dirname "$log")"
timeout -k 10 "$lim" "$@" > "$log" 2>&1
rc=$?
echo "[gpu_step] rc $rc: $*" >> "$log"
echo "[gpu_step] rc $rc: $* (log $log)"
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
exit 0
