#!/bin/bash
# gpurun with retries while the pool is busy (exit code 3 = nothing charged): tools/grun.sh <timeout-seconds> '<command>'
t=$1; shift
for i in $(seq 1 20); do
  /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 60
done
exit 3
