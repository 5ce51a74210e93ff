#!/bin/bash
# round 6: AdamW riders (MB_ADAMW_RIDE=1) -- parity, then in-step A/B by tile of the grouped weight gradient
mkdir -p gpurun_out/r06m
O=gpurun_out/r06m/ride_ab.txt
: > $O
(timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider -k "riding or epilogue_changes" 2>&1 | tail -15) >> $O 2>&1
ARGS="--steps 200 --warmup 30 --graph 1 --h2d 2"
for rep in 1 2; do
  for cfg in "MB_GROUP_WGRAD=128 MB_ADAMW_RIDE=0" "MB_GROUP_WGRAD=128 MB_ADAMW_RIDE=1" "MB_GROUP_WGRAD=256 MB_ADAMW_RIDE=1" "MB_GROUP_WGRAD=128 MB_ADAMW_IN_WGRAD=1"; do
    echo "== $cfg" >> $O
    env $cfg timeout 120 tools/bin/step_bench $ARGS 2>&1 | grep "ms/step" >> $O
  done
done
echo "== C5 shape" >> $O
for cfg in "MB_GROUP_WGRAD=128 MB_ADAMW_RIDE=0" "MB_GROUP_WGRAD=128 MB_ADAMW_RIDE=1" "MB_GROUP_WGRAD=256 MB_ADAMW_RIDE=1"; do
    echo "== $cfg" >> $O
    env $cfg timeout 120 tools/bin/step_bench $ARGS --batch 32 --seq 128 --visual 35 2>&1 | grep "ms/step" >> $O
done
cat $O
