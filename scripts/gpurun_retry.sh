#!/bin/bash
# gpurun with retries while every GPU slot of the pod is busy (exit code 3 = nothing charged).  usage: gpurun_retry.sh <timeout> <logfile> <command string>
to=$1; log=$2; shift; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $to -- "$@" > $log 2>&1
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 45
done
exit 3
