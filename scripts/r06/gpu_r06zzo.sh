#!/bin/bash
# MAG-XLNet: attention-host rider budget above 3 M per launch
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
OUT=$R/gpurun_out/r06_xlnet_ride_budget2.txt
SB=$R/tools/bin/step_bench
bash scripts/box_log.sh > /dev/null 2>&1
{
for rep in 1 2 3; do
  for cfg in "MB_X=0" "MB_ADAMW_RIDE_ATTN_PARAMS=3600000" "MB_ADAMW_RIDE_ATTN_PARAMS=4200000" "MB_ADAMW_RIDE_ATTN_PARAMS=4800000" "MB_ADAMW_RIDE_ATTN_PARAMS=5400000"; do
    echo "== step xlnet $cfg"; env $cfg timeout 60 $SB --model xlnet --graph 1 --h2d 2 --steps 60 --warmup 10 2>&1 | grep -o "[0-9.]* ms/step (events)"
  done
done
MB_GEMM_LOG=1 timeout 60 $SB --model xlnet --graph 1 --h2d 2 --steps 3 --warmup 1 2>&1 | grep -E "magbert ride|magbert adamw" | sort | uniq -c | sort -rn | head
} > $OUT 2>&1
cat $OUT
