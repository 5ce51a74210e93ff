#!/bin/bash
# new defaults (attention rider budget 1,650 / token, MAG-XLNet grouped weight gradient on the ping-pong tile): parity subset + timings + XLNet attention budget
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
OUT=$R/gpurun_out/r06_final_defaults.txt
SB=$R/tools/bin/step_bench
export TMPDIR=/tmp
bash scripts/box_log.sh > /dev/null 2>&1
{
timeout 1200 python -m pytest tests/test_xlnet_gpu.py tests/test_model_gpu.py -q -x -k "xlnet or riders or riding or graph" 2>&1 | tail -4
for rep in 1 2 3; do
  echo "== step B=48 L=50"; timeout 60 $SB --graph 1 --h2d 2 --steps 100 --warmup 10 2>&1 | grep -o "[0-9.]* ms/step (events)"
  echo "== step C5"; timeout 60 $SB --graph 1 --h2d 2 --batch 32 --seq 128 --visual 35 --steps 60 --warmup 10 2>&1 | grep -o "[0-9.]* ms/step (events)"
  for cfg in "MB_X=0" "MB_ADAMW_RIDE_ATTN_PARAMS=2400000" "MB_ADAMW_RIDE_ATTN_PARAMS=3000000"; do
    echo "== step xlnet $cfg"; env $cfg timeout 60 $SB --model xlnet --graph 1 --h2d 2 --steps 60 --warmup 10 2>&1 | grep -o "[0-9.]* ms/step (events)"
  done
done
} > $OUT 2>&1
cat $OUT
