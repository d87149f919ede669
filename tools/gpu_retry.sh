#!/bin/bash
# gpurun with retries while the pod's GPU slots are busy (exit code 3 = nothing charged): tools/gpu_retry.sh <timeout-seconds> '<command>'
t=$1; shift
for i in $(seq 1 20); do
  /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 90
done
exit 3
