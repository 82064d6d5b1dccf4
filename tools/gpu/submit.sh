#!/bin/bash
# gpurun with patience: resubmits ONLY while the pod reports "no box or slot free" (exit code 3: nothing ran, nothing was
# charged).  Any other outcome -- success, a failing command, a timeout, a refusal -- is returned at once, never retried.
#   tools/gpu/submit.sh <timeout-seconds> '<command>'
T=$1; shift
for try in $(seq 1 30); do
  /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 90
done
exit 3
