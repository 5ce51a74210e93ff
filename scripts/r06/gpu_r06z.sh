#!/bin/bash
# AdamW riders in the 64 x 64 dgrad launches (MB_ADAMW_RIDE_DGRAD): parity test, then same-box A/B over the budget per launch
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
bash scripts/box_log.sh > /dev/null 2>&1
SB=$R/tools/bin/step_bench
timeout 600 python -m pytest tests/test_model_gpu.py -x -q -k "riding" 2>&1 | tail -3
O=$R/gpurun_out/r06_adamw_ride_dgrad.txt
{
for rep in 1 2; do
  for cfg in "MB_ADAMW_RIDE_DGRAD=0" "MB_ADAMW_RIDE_DGRAD=1" "MB_ADAMW_RIDE_DGRAD=1 MB_ADAMW_RIDE_DGRAD_PARAMS=1000000" "MB_ADAMW_RIDE_DGRAD=1 MB_ADAMW_RIDE_DGRAD_PARAMS=1500000" "MB_ADAMW_RIDE_DGRAD=1 MB_ADAMW_RIDE_DGRAD_PARAMS=2500000" "MB_ADAMW_RIDE_DGRAD=1 MB_ADAMW_RIDE_DGRAD_PARAMS=3500000" "MB_ADAMW_RIDE_DGRAD=1 MB_ADAMW_RIDE_DGRAD_PARAMS=2000000 MB_ADAMW_RIDE_PARAMS=1024" "MB_ADAMW_RIDE_DGRAD=1 MB_ADAMW_RIDE_DGRAD_PARAMS=3000000 MB_ADAMW_RIDE_PARAMS=1024"; do
    echo "== $cfg"
    env $cfg timeout 60 $SB --graph 1 --h2d 2 --steps 300 --warmup 20 2>&1 | grep -o "[0-9.]* ms/step (events)"
  done
done
} > $O 2>&1
cat $O
