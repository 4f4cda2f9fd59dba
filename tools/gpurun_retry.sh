#!/bin/bash
# usage: tools/gpurun_retry.sh TIMEOUT 'command'   — retries while the pool has no free slot (nothing is charged for those attempts)
T=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@" > /tmp/gpurun_last.log 2>&1
  rc=$?
  if grep -q "status=transient" /tmp/gpurun_last.log; then sleep 90; continue; fi
  break
done
cat /tmp/gpurun_last.log
exit $rc
