#!/bin/bash
# MB_ADAMW_OVERLAP=C: optimizer of finished layer chunks on a side stream under the backward of the layers below (same-box A/B)
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out; export TMPDIR=/tmp
OUT=gpurun_out/r3_opt_overlap.txt
: > $OUT
for rep in 1 2; do
for c in 0 4 2 6 12 1; do
  echo "== MB_ADAMW_OVERLAP=$c graph" >> $OUT
  MB_ADAMW_OVERLAP=$c timeout 120 tools/bin/step_bench --graph 1 --h2d 2 --steps 200 --warmup 30 2>&1 | tail -1 >> $OUT
done
done
for c in 0 4; do
  echo "== MB_ADAMW_OVERLAP=$c launches" >> $OUT
  MB_ADAMW_OVERLAP=$c timeout 120 tools/bin/step_bench --graph 0 --h2d 2 --steps 200 --warmup 30 2>&1 | tail -1 >> $OUT
done
echo "== tests under MB_ADAMW_OVERLAP=4" >> $OUT
MB_ADAMW_OVERLAP=4 timeout 600 python -m pytest tests/test_model_gpu.py -q -x -k "step_graph or single_call or known_zero or fused" 2>&1 | tail -5 >> $OUT
cat $OUT
