#!/bin/bash
# rider allocation between the two big hosts of a layer (everything that is final is taken either way): weight gradient vs attention backward
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
OUT=$R/gpurun_out/r06_ride_alloc.txt
SB=$R/tools/bin/step_bench
bash scripts/box_log.sh > /dev/null 2>&1
{
for rep in 1 2 3; do
  for cfg in "MB_X=0" "MB_ADAMW_RIDE_PARAMS=2000000 MB_ADAMW_RIDE_ATTN_PARAMS=4500000" "MB_ADAMW_RIDE_PARAMS=1500000 MB_ADAMW_RIDE_ATTN_PARAMS=5000000" "MB_ADAMW_RIDE_PARAMS=1000000 MB_ADAMW_RIDE_ATTN_PARAMS=5500000" "MB_ADAMW_RIDE_PARAMS=2800000 MB_ADAMW_RIDE_ATTN_PARAMS=3600000"; do
    echo "== step B=48 L=50 $cfg"; env $cfg timeout 60 $SB --graph 1 --h2d 2 --steps 100 --warmup 10 2>&1 | grep -o "[0-9.]* ms/step (events)"
  done
done
} > $OUT 2>&1
cat $OUT
