#!/bin/bash
# MAG-XLNet riders (dgrad hosts): parity tests, then same-box A/B
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
bash scripts/box_log.sh > /dev/null 2>&1
timeout 900 python -m pytest tests/test_xlnet_gpu.py tests/test_model_gpu.py -x -q -k "riders or riding or single_call_step" 2>&1 | tail -3
SB=$R/tools/bin/step_bench
for rep in 1 2 3; do
  for cfg in "MB_ADAMW_RIDE=0" "MB_ADAMW_RIDE=1" "MB_ADAMW_RIDE=1 MB_ADAMW_RIDE_DGRAD=1" "MB_ADAMW_RIDE=1 MB_ADAMW_RIDE_DGRAD_PARAMS=2000000"; do
    echo "== xlnet $cfg"; env $cfg timeout 60 $SB --model xlnet --graph 1 --h2d 2 --steps 150 --warmup 10 2>&1 | grep -o "[0-9.]* ms/step (events)"
  done
done
MB_GEMM_LOG=1 timeout 60 $SB --model xlnet --graph 1 --h2d 2 --steps 3 --warmup 1 2>&1 | grep -E "magbert ride|magbert adamw" | sort | uniq -c | sort -rn | head
