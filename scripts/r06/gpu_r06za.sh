#!/bin/bash
# full GPU suite + dgrad-rider block-count A/B
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
bash scripts/box_log.sh > /dev/null 2>&1
SB=$R/tools/bin/step_bench
O=$R/gpurun_out/r06_adamw_ride_dgrad_blocks.txt
{
for rep in 1 2; do
  for cfg in "MB_ADAMW_RIDE_DGRAD=0" "MB_ADAMW_RIDE_DGRAD=1" "MB_ADAMW_RIDE_DGRAD_BLOCKS=64" "MB_ADAMW_RIDE_DGRAD_BLOCKS=128" "MB_ADAMW_RIDE_DGRAD_BLOCKS=256" "MB_ADAMW_RIDE_DGRAD_BLOCKS=128 MB_ADAMW_RIDE_DGRAD_PARAMS=2000000"; do
    echo "== $cfg"
    env $cfg timeout 60 $SB --graph 1 --h2d 2 --steps 300 --warmup 20 2>&1 | grep -o "[0-9.]* ms/step (events)"
  done
done
echo "== C5"
for cfg in "MB_ADAMW_RIDE_DGRAD=0" "MB_ADAMW_RIDE_DGRAD=1"; do echo "== $cfg"; env $cfg timeout 60 $SB --graph 1 --h2d 2 --batch 32 --seq 128 --visual 35 --steps 100 --warmup 10 2>&1 | grep -o "[0-9.]* ms/step (events)"; done
} > $O 2>&1
cat $O
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 | tee $R/gpurun_out/r06_gpu_suite_tail.txt
