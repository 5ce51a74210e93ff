#!/bin/bash
# round 6: AdamW riders with a per-launch parameter budget
mkdir -p gpurun_out/r06n
O=gpurun_out/r06n/ride_ab.txt
: > $O
(timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider -k "riding" 2>&1 | tail -15) >> $O 2>&1
ARGS="--steps 200 --warmup 30 --graph 1 --h2d 2"
for rep in 1 2; do
  for cfg in "MB_GROUP_WGRAD=128 MB_ADAMW_RIDE=0" "MB_GROUP_WGRAD=256 MB_ADAMW_RIDE=0" "MB_GROUP_WGRAD=256 MB_ADAMW_RIDE=1 MB_ADAMW_RIDE_PARAMS=1000000" "MB_GROUP_WGRAD=256 MB_ADAMW_RIDE=1 MB_ADAMW_RIDE_PARAMS=2000000" "MB_GROUP_WGRAD=256 MB_ADAMW_RIDE=1 MB_ADAMW_RIDE_PARAMS=3000000" "MB_GROUP_WGRAD=256 MB_ADAMW_RIDE=1 MB_ADAMW_RIDE_PARAMS=4000000" "MB_GROUP_WGRAD=128 MB_ADAMW_RIDE=1 MB_ADAMW_RIDE_PARAMS=2000000" "MB_GROUP_WGRAD=128 MB_ADAMW_RIDE=1 MB_ADAMW_RIDE_PARAMS=4000000"; do
    echo "== $cfg" >> $O
    env $cfg timeout 120 tools/bin/step_bench $ARGS 2>&1 | grep "ms/step" | cut -c1-140 >> $O
  done
done
cat $O
