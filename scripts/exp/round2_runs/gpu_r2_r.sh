#!/bin/bash
# known-zero gradient stores: full GPU suite + A/B of the step with and without them
mkdir -p gpurun_out
out=gpurun_out/overwrite.txt
: > $out
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -12 >> $out
for rep in 1 2; do
for v in 1 0; do echo "== MB_WGRAD_OVERWRITE=$v" >> $out; MB_WGRAD_OVERWRITE=$v timeout 120 tools/bin/step_bench --steps 200 --warmup 30 --graph 1 --h2d 2 >> $out 2>&1; done
done
MB_WGRAD_OVERWRITE=1 timeout 120 tools/bin/step_bench --steps 100 --warmup 20 --graph 1 --h2d 2 --batch 32 --seq 128 --visual 35 >> $out 2>&1
cat $out
