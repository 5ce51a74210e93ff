#!/bin/bash
# rider budget of the grouped weight gradient now that its launch is shorter; 256 x 128 tiles for the forward wide launches only
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
OUT=$R/gpurun_out/r06_ride_budget2.txt
SB=$R/tools/bin/step_bench
bash scripts/box_log.sh > /dev/null 2>&1
{
for rep in 1 2 3; do
  for cfg in "MB_X=0" "MB_ADAMW_RIDE_PARAMS=1600000" "MB_ADAMW_RIDE_PARAMS=1900000" "MB_ADAMW_RIDE_PARAMS=2200000" "MB_GEMM_TILE_BIG=4" "MB_ADAMW_RIDE_ATTN_PARAMS=2000000" "MB_ADAMW_RIDE_ATTN_PARAMS=3000000"; do
    echo "== step B=48 L=50 $cfg"; env $cfg timeout 60 $SB --graph 1 --h2d 2 --steps 100 --warmup 10 2>&1 | grep -o "[0-9.]* ms/step (events)"
  done
done
for rep in 1 2; do
  for cfg in "MB_X=0" "MB_ADAMW_RIDE_PARAMS=3300000" "MB_ADAMW_RIDE_PARAMS=3800000"  "MB_GEMM_TILE_BIG=4"; do
    echo "== step C5 $cfg"; env $cfg timeout 60 $SB --graph 1 --h2d 2 --batch 32 --seq 128 --visual 35 --steps 60 --warmup 10 2>&1 | grep -o "[0-9.]* ms/step (events)"
  done
done
MB_GEMM_LOG=1 timeout 60 $SB --graph 1 --h2d 2 --batch 32 --seq 128 --visual 35 --steps 3 --warmup 1 2>&1 | grep -E "magbert ride|magbert adamw" | sort | uniq -c | sort -rn | head
} > $OUT 2>&1
cat $OUT
