#!/bin/bash
# gpurun with retries while the pod's GPU slots are busy (exit code 3 / "transient": nothing charged).   tools/gpurun_retry.sh TIMEOUT 'command'
T=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@" > /tmp/gpurun_retry.$$ 2>&1; rc=$?
  if grep -q "status=transient" /tmp/gpurun_retry.$$; then sleep 90; continue; fi
  grep -v "^\[gpurun\] send" /tmp/gpurun_retry.$$; exit $rc
done
echo "gave up: slots busy"; exit 3
