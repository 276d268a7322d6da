#!/bin/bash
# gpurun with retries while the pod's GPU slots are busy (exit code 3 = nothing charged):  tools/grun.sh <timeout_s> '<command>'
t=$1; shift
for i in $(seq 40); do
  /usr/local/graft/bin/gpurun --timeout $t -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 45
done
exit 3
