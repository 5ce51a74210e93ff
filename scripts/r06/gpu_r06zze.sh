#!/bin/bash
# attention-backward rider budget scan (the kernels around it got shorter); MAG-XLNet grouped weight gradient on the ping-pong tile again
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
OUT=$R/gpurun_out/r06_ride_budget3.txt
SB=$R/tools/bin/step_bench
bash scripts/box_log.sh > /dev/null 2>&1
{
for rep in 1 2 3; do
  for cfg in "MB_X=0" "MB_ADAMW_RIDE_ATTN_PARAMS=3000000" "MB_ADAMW_RIDE_ATTN_PARAMS=3500000" "MB_ADAMW_RIDE_ATTN_PARAMS=4000000" "MB_ADAMW_RIDE_ATTN_PARAMS=4500000"; do
    echo "== step B=48 L=50 $cfg"; env $cfg timeout 60 $SB --graph 1 --h2d 2 --steps 100 --warmup 10 2>&1 | grep -o "[0-9.]* ms/step (events)"
  done
  for cfg in "MB_X=0" "MB_GROUP_WGRAD=256"; do
    echo "== step xlnet $cfg"; env $cfg timeout 60 $SB --model xlnet --graph 1 --h2d 2 --steps 60 --warmup 10 2>&1 | grep -o "[0-9.]* ms/step (events)"
  done
done
MB_GEMM_LOG=1 timeout 60 $SB --graph 1 --h2d 2 --steps 3 --warmup 1 2>&1 | grep -E "magbert ride|magbert adamw" | sort | uniq -c | sort -rn | head
} > $OUT 2>&1
cat $OUT
