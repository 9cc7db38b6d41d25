#!/bin/bash
# gpurun with retries while every GPU slot of the pod is busy (exit code 3: nothing charged).  Usage: tools/gpurun_retry.sh <timeout_s> '<command>'
t=$1; shift
for i in $(seq 1 30); do
  /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 60
done
exit 3
