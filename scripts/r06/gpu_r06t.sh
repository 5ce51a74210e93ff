#!/bin/bash
# round 6: careful same-box A/B of the two new defaults (alternating, 4 repetitions, 300 timed steps)
mkdir -p gpurun_out/r06t
O=gpurun_out/r06t/defaults_ab.txt
: > $O
rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|mclk\|power" | head -6 >> $O
ARGS="--steps 300 --warmup 30 --graph 1 --h2d 2"
for rep in 1 2 3 4; do
  for cfg in "MB_GROUP_WGRAD=128 MB_ADAMW_RIDE=0" "MB_GROUP_WGRAD=256 MB_ADAMW_RIDE=0" "MB_GROUP_WGRAD=256 MB_ADAMW_RIDE=1" "MB_GROUP_WGRAD=128 MB_ADAMW_RIDE=1" "MB_GROUP_WGRAD=128 MB_ADAMW_IN_WGRAD=1"; do
    echo "== $cfg" >> $O
    env $cfg timeout 120 tools/bin/step_bench $ARGS 2>&1 | grep "ms/step" | cut -c64-110 >> $O
  done
done
echo "== C5" >> $O
for rep in 1 2 3; do
  for cfg in "MB_GROUP_WGRAD=128 MB_ADAMW_RIDE=0" "MB_GROUP_WGRAD=256 MB_ADAMW_RIDE=1"; do
    echo "== $cfg" >> $O
    env $cfg timeout 120 tools/bin/step_bench $ARGS --batch 32 --seq 128 --visual 35 2>&1 | grep "ms/step" | cut -c64-110 >> $O
  done
done
paste - - < $O | sed 's/step_bench//' | sort | head -60
