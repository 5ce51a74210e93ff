#!/bin/bash
# round 6: in-step A/B of the grouped weight gradient tile (MB_GROUP_WGRAD=128 | 256) and of the big forward / dgrad tile (MB_GEMM_TILE_BIG)
mkdir -p gpurun_out/r06k
O=gpurun_out/r06k/step_ab.txt
: > $O
ARGS="--steps 200 --warmup 30 --graph 1 --h2d 2"
for rep in 1 2 3; do
  for cfg in "MB_GROUP_WGRAD=128" "MB_GROUP_WGRAD=256" "MB_GROUP_WGRAD=256 MB_GEMM_TILE_BIG=1"; do
    echo "== $cfg" >> $O
    env $cfg timeout 120 tools/bin/step_bench $ARGS 2>&1 | grep "ms/step" >> $O
  done
done
echo "== C5 shape" >> $O
for rep in 1 2; do
  for cfg in "MB_GROUP_WGRAD=128" "MB_GROUP_WGRAD=256" "MB_GROUP_WGRAD=256 MB_GEMM_TILE_BIG=2"; do
    echo "== $cfg" >> $O
    env $cfg timeout 120 tools/bin/step_bench $ARGS --batch 32 --seq 128 --visual 35 2>&1 | grep "ms/step" >> $O
  done
done
cat $O
