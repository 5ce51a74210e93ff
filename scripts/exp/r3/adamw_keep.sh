#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out; out=gpurun_out/r3_adamw_keep.txt; : > $out
S="tools/bin/step_bench --steps 200 --warmup 30 --graph 1 --h2d 2"
for rep in 1 2; do
  echo "== default (VAR=3, keep)" >> $out; timeout 120 $S >> $out 2>&1
  echo "== MB_ADAMW_KEEP=0" >> $out; MB_ADAMW_KEEP=0 timeout 120 $S >> $out 2>&1
  echo "== MB_ADAMW_VAR=0 MB_ADAMW_KEEP=0" >> $out; MB_ADAMW_VAR=0 MB_ADAMW_KEEP=0 timeout 120 $S >> $out 2>&1
done
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 >> $out
cat $out
