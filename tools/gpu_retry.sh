#!/bin/bash
# usage: gpu_retry.sh <logfile> <timeout> <command...>
LOG=$1; TO=$2; shift 2
for i in $(seq 1 40); do
  timeout $((TO+1200)) gpurun --timeout $TO -- "$@" > $LOG 2>&1
  if ! grep -q "status=transient" $LOG; then exit 0; fi
  sleep 60
done
