#!/bin/bash
# rider workgroups of the ping-pong grouped launch: more quads in flight per thread (MB_RIDE_UNR builds) x parameters per launch
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
bash scripts/box_log.sh > /dev/null 2>&1
SB=$R/tools/bin/step_bench
for rep in 1 2; do
for v in base unr4 unr5 unr6; do
  for pr in 0 3000000 3600000 4200000; do
    L=$LD_LIBRARY_PATH; [ $v != base ] && L=$R/gpurun_ab/$v:$LD_LIBRARY_PATH
    echo "== $v MB_ADAMW_RIDE_PARAMS=$pr"
    [ $pr = 0 ] && LD_LIBRARY_PATH=$L timeout 60 $SB --graph 1 --h2d 2 --steps 300 --warmup 20 2>&1 | grep -o "[0-9.]* ms/step (events)"
    [ $pr != 0 ] && LD_LIBRARY_PATH=$L MB_ADAMW_RIDE_PARAMS=$pr timeout 60 $SB --graph 1 --h2d 2 --steps 300 --warmup 20 2>&1 | grep -o "[0-9.]* ms/step (events)"
  done
done
done
