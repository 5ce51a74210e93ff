#!/bin/bash
# AdamW riders in the attention backward launch (MB_ADAMW_RIDE_ATTN): parity, then same-box A/B over the budget, both shapes
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
bash scripts/box_log.sh > /dev/null 2>&1
timeout 600 python -m pytest tests/test_model_gpu.py tests/test_ops_gpu.py -x -q -k "riding or attention" 2>&1 | tail -2
SB=$R/tools/bin/step_bench
for rep in 1 2 3; do
  for cfg in "MB_ADAMW_RIDE_ATTN=0" "MB_ADAMW_RIDE_ATTN=1" "MB_ADAMW_RIDE_ATTN_PARAMS=500000" "MB_ADAMW_RIDE_ATTN_PARAMS=1500000" "MB_ADAMW_RIDE_ATTN_PARAMS=1000000 MB_ADAMW_RIDE_ATTN_BLOCKS=256"; do
    echo "== B=48 L=50 $cfg"; env $cfg timeout 60 $SB --graph 1 --h2d 2 --steps 300 --warmup 20 2>&1 | grep -o "[0-9.]* ms/step (events)"
  done
done
for rep in 1 2 3; do
  for cfg in "MB_ADAMW_RIDE_ATTN=0" "MB_ADAMW_RIDE_ATTN=1" "MB_ADAMW_RIDE_ATTN_PARAMS=1500000" "MB_ADAMW_RIDE_ATTN_PARAMS=3500000" "MB_ADAMW_RIDE_ATTN_PARAMS=5000000"; do
    echo "== B=32 L=128 $cfg"; env $cfg timeout 60 $SB --graph 1 --h2d 2 --batch 32 --seq 128 --visual 35 --steps 100 --warmup 10 2>&1 | grep -o "[0-9.]* ms/step (events)"
  done
done
