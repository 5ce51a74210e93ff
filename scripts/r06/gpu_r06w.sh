#!/bin/bash
mkdir -p gpurun_out/r06w
O=gpurun_out/r06w/xlnet_wgrad_tile.txt
: > $O
ARGS="--model xlnet --steps 200 --warmup 30 --graph 1 --h2d 2"
for rep in 1 2 3; do
  for cfg in "MB_GROUP_WGRAD=128" "MB_GROUP_WGRAD=256"; do
    echo "== $cfg" >> $O
    env $cfg timeout 120 tools/bin/step_bench $ARGS 2>&1 | grep "ms/step" | cut -c1-150 >> $O
  done
done
cat $O
