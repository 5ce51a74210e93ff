#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r04c; mkdir -p $O; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
SB=$R/tools/bin/step_bench
A="--graph 1 --h2d 2 --steps 100 --warmup 20"
{
for rep in 1 2 3; do
  echo "== single"; timeout 120 $SB $A
  for m in 0 1 2 3; do
    echo "== dp event_mode $m"; MB_DP_EVENT_MODE=$m timeout 180 $SB $A --dp 1 2>&1 | grep "ms/step"
  done
  echo "== dp debug1"; MB_DP_DEBUG=1 timeout 180 $SB $A --dp 1 2>&1 | grep "ms/step"
done
} > $O/dp_event_modes.txt 2>&1
cat $O/dp_event_modes.txt | cut -c1-160
