#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$(pwd); OUT=gpurun_out/r3_attn128.txt; : > $OUT
for rep in 1 2; do
echo "== base" >> $OUT; LD_LIBRARY_PATH=$R/gpurun_ab/base timeout 60 tools/bin/attn_bench --batch 32 --seq 128 2>&1 | grep "us/launch" >> $OUT
echo "== current (backward capped at 128 VGPRs, 118 spilled)" >> $OUT; timeout 60 tools/bin/attn_bench --batch 32 --seq 128 2>&1 | grep "us/launch" >> $OUT
done
S="tools/bin/step_bench --steps 150 --warmup 20 --graph 1 --h2d 2 --batch 32 --seq 128 --visual 35"
echo "== current C5" >> $OUT; timeout 120 $S 2>&1 | tail -1 >> $OUT
timeout 300 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "attention" 2>&1 | tail -2 >> $OUT
cat $OUT
