#!/bin/bash
# Build-container helper: submit one gpurun call, retrying only while the pod reports "no slot free" (exit code 3, nothing
# charged).  usage: tools/gpurun_retry.sh <timeout-seconds> '<command>'
T=$1; shift
for i in $(seq 1 20); do
  /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 60
done
exit 3
