#!/bin/bash
# round 6: AdamW riders in the LayerNorm-backward launches (MB_ADAMW_RIDE_LN = parameters per launch)
mkdir -p gpurun_out/r06s
O=gpurun_out/r06s/ride_ln.txt
: > $O
(timeout 900 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider -k "riding" 2>&1 | tail -4) >> $O 2>&1
(MB_ADAMW_RIDE_LN=300000 timeout 900 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider -k "riding" 2>&1 | tail -4) >> $O 2>&1
ARGS="--steps 200 --warmup 30 --graph 1 --h2d 2"
for rep in 1 2; do
  for cfg in "MB_ADAMW_RIDE=0" "MB_ADAMW_RIDE_LN=0" "MB_ADAMW_RIDE_LN=500000" "MB_ADAMW_RIDE_LN=1000000" "MB_ADAMW_RIDE_LN=1500000" "MB_ADAMW_RIDE_LN=2500000" "MB_ADAMW_RIDE_LN=1000000 MB_ADAMW_RIDE_PARAMS=1024"; do
    echo "== $cfg" >> $O
    env $cfg timeout 120 tools/bin/step_bench $ARGS 2>&1 | grep "ms/step" | cut -c1-140 >> $O
  done
done
cat $O
