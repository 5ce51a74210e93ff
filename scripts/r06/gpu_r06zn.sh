#!/bin/bash
# attention-backward riders at L = 50: budget sweep upward
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
bash scripts/box_log.sh > /dev/null 2>&1
SB=$R/tools/bin/step_bench
for rep in 1 2 3; do
  for cfg in "MB_ADAMW_RIDE_ATTN=0" "MB_ADAMW_RIDE_ATTN_PARAMS=1500000" "MB_ADAMW_RIDE_ATTN_PARAMS=2000000" "MB_ADAMW_RIDE_ATTN_PARAMS=2500000" "MB_ADAMW_RIDE_ATTN_PARAMS=3000000" "MB_ADAMW_RIDE_ATTN_PARAMS=2000000 MB_ADAMW_RIDE_PARAMS=1800000" "MB_ADAMW_RIDE_ATTN_PARAMS=2000000 MB_ADAMW_RIDE_DGRAD=0"; do
    echo "== B=48 L=50 $cfg"; env $cfg timeout 60 $SB --graph 1 --h2d 2 --steps 300 --warmup 20 2>&1 | grep -o "[0-9.]* ms/step (events)"
  done
done
