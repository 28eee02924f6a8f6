#!/bin/bash
# usage: tools/grun.sh <tag> <timeout_s> '<command>'   -- runs one gpurun call, retrying while no slot is free; log in gpurun_out/<tag>.call
tag=$1; to=$2; shift 2
while true; do
  while /usr/local/graft/bin/gpurun --status | grep -q '"in_flight": 1'; do sleep 5; done
  /usr/local/graft/bin/gpurun --timeout "$to" -- "$@" > gpurun_out/$tag.call 2>&1
  if grep -q "status=transient\|status=refused" gpurun_out/$tag.call; then sleep 45; else break; fi
done
