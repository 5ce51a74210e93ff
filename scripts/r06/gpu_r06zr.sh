#!/bin/bash
# rebalancing the hosts now that a layer is ridden completely: weight-gradient rider budget x attention budget
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
bash scripts/box_log.sh > /dev/null 2>&1
SB=$R/tools/bin/step_bench
for rep in 1 2 3; do
  for cfg in "X=0" "MB_ADAMW_RIDE_PARAMS=1900000" "MB_ADAMW_RIDE_PARAMS=1500000" "MB_ADAMW_RIDE_PARAMS=1900000 MB_ADAMW_RIDE_ATTN_PARAMS=3000000" "MB_ADAMW_RIDE_PARAMS=1500000 MB_ADAMW_RIDE_ATTN_PARAMS=3000000" "MB_ADAMW_RIDE_PARAMS=1000000 MB_ADAMW_RIDE_ATTN_PARAMS=3000000 MB_ADAMW_RIDE_DGRAD_PARAMS=1600000"; do
    echo "== $cfg"; env $cfg timeout 60 $SB --graph 1 --h2d 2 --steps 300 --warmup 20 2>&1 | grep -o "[0-9.]* ms/step (events)"
  done
done
