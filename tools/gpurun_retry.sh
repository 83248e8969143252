#!/bin/bash
# gpurun with retries while every GPU slot of the pod is busy (exit code 3: nothing charged).  usage: gpurun_retry.sh <timeout s> <command>
T=$1; shift
for i in $(seq 1 12); do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 120
done
exit 3
