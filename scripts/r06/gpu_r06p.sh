#!/bin/bash
mkdir -p gpurun_out/r06p
O=gpurun_out/r06p/ride_ab.txt
: > $O
(timeout 900 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider -k "riding or epilogue_changes or adamw or optim" 2>&1 | tail -8) >> $O 2>&1
ARGS="--steps 200 --warmup 30 --graph 1 --h2d 2"
for rep in 1 2; do
  for cfg in "MB_GROUP_WGRAD=128 MB_ADAMW_RIDE=0" "MB_GROUP_WGRAD=256 MB_ADAMW_RIDE=1 MB_ADAMW_RIDE_PARAMS=2500000" "MB_GROUP_WGRAD=256 MB_ADAMW_RIDE=1 MB_ADAMW_RIDE_PARAMS=3500000" "MB_GROUP_WGRAD=256 MB_ADAMW_RIDE=1 MB_ADAMW_RIDE_PARAMS=4500000" "MB_GROUP_WGRAD=256 MB_ADAMW_RIDE=1 MB_ADAMW_RIDE_PARAMS=6000000" "MB_GROUP_WGRAD=128 MB_ADAMW_RIDE=1 MB_ADAMW_RIDE_PARAMS=3500000"; do
    echo "== $cfg" >> $O
    env $cfg timeout 120 tools/bin/step_bench $ARGS 2>&1 | grep "ms/step" | cut -c1-140 >> $O
  done
done
echo "== C5" >> $O
for cfg in "MB_GROUP_WGRAD=128 MB_ADAMW_RIDE=0" "MB_GROUP_WGRAD=256 MB_ADAMW_RIDE=0" "MB_GROUP_WGRAD=256 MB_ADAMW_RIDE=1 MB_ADAMW_RIDE_PARAMS=2000000" "MB_GROUP_WGRAD=256 MB_ADAMW_RIDE=1 MB_ADAMW_RIDE_PARAMS=3500000" "MB_GROUP_WGRAD=256 MB_ADAMW_RIDE=1 MB_ADAMW_RIDE_PARAMS=5000000"; do
    echo "== $cfg" >> $O
    env $cfg timeout 120 tools/bin/step_bench $ARGS --batch 32 --seq 128 --visual 35 2>&1 | grep "ms/step" | cut -c1-140 >> $O
done
cat $O
