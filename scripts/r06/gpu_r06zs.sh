#!/bin/bash
# C5 (B=32 L=128, T=4096): riders in the second round of the 128 x 128 ffn2 dgrad (768 tiles in 512 slots)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
bash scripts/box_log.sh > /dev/null 2>&1
SB=$R/tools/bin/step_bench
for rep in 1 2 3; do
  for cfg in "MB_ADAMW_RIDE_DGRAD=1" "MB_ADAMW_RIDE_DGRAD=2" "MB_ADAMW_RIDE_DGRAD=2 MB_ADAMW_RIDE_DGELU_PARAMS=2000000" "MB_ADAMW_RIDE_DGRAD=2 MB_ADAMW_RIDE_DGELU_PARAMS=5000000"; do
    echo "== B=32 L=128 $cfg"; env $cfg timeout 60 $SB --graph 1 --h2d 2 --batch 32 --seq 128 --visual 35 --steps 100 --warmup 10 2>&1 | grep -o "[0-9.]* ms/step (events)"
  done
done
MB_GEMM_LOG=1 timeout 60 $SB --graph 1 --h2d 2 --batch 32 --seq 128 --visual 35 --steps 3 --warmup 1 2>&1 | grep -E "magbert ride|magbert adamw" | sort | uniq -c | sort -rn | head
