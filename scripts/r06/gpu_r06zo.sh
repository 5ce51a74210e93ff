#!/bin/bash
# new defaults: parity, A/B, per-launch ride log
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
bash scripts/box_log.sh > /dev/null 2>&1
timeout 600 python -m pytest tests/test_model_gpu.py -x -q -k "riding" 2>&1 | tail -1
SB=$R/tools/bin/step_bench
for rep in 1 2 3; do
  for cfg in "MB_ADAMW_RIDE_ATTN=0" "MB_ADAMW_RIDE_ATTN=1" "MB_ADAMW_RIDE_ATTN=1 MB_ADAMW_RIDE_DGRAD=1" "MB_ADAMW_RIDE_ATTN=1 MB_ADAMW_RIDE_DGRAD_PARAMS=600000"; do
    echo "== B=48 L=50 $cfg"; env $cfg timeout 60 $SB --graph 1 --h2d 2 --steps 300 --warmup 20 2>&1 | grep -o "[0-9.]* ms/step (events)"
  done
done
MB_GEMM_LOG=1 timeout 60 $SB --graph 1 --h2d 2 --steps 3 --warmup 1 2>&1 | grep -E "magbert ride|magbert adamw" | tail -60 | sort | uniq -c | sort -rn | head -20
