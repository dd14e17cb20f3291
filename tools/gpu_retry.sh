#!/bin/bash
# gpurun with retries while the pod's GPU slots are busy (exit code 3 = nothing charged): tools/gpu_retry.sh <timeout> <command>
T=$1; shift
for i in $(seq 1 30); do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 45
done
exit 3
