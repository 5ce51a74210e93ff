#!/bin/bash
# attention kernels with the probabilities kept in registers (AccOp): parity tests, attn_bench, same-box step A/B vs gpurun_ab/base
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$(pwd); OUT=gpurun_out/r3_attn_regs.txt; : > $OUT
timeout 600 python -m pytest tests -x -q -m gpu -k "attention or attn or head_mask or dropout or oracle" 2>&1 | tail -4 >> $OUT
echo "== attn_bench base" >> $OUT; LD_LIBRARY_PATH=$R/gpurun_ab/base timeout 120 tools/bin/attn_bench 2>&1 | grep "us/launch" >> $OUT
echo "== attn_bench current" >> $OUT; timeout 120 tools/bin/attn_bench 2>&1 | grep "us/launch" >> $OUT
S="tools/bin/step_bench --steps 200 --warmup 30 --graph 1 --h2d 2"
for rep in 1 2; do
  echo "== base" >> $OUT; LD_LIBRARY_PATH=$R/gpurun_ab/base timeout 120 $S 2>&1 | tail -1 >> $OUT
  echo "== current" >> $OUT; timeout 120 $S 2>&1 | tail -1 >> $OUT
done
echo "== base C5" >> $OUT; LD_LIBRARY_PATH=$R/gpurun_ab/base timeout 120 $S --batch 32 --seq 128 --visual 35 2>&1 | tail -1 >> $OUT
echo "== current C5" >> $OUT; timeout 120 $S --batch 32 --seq 128 --visual 35 2>&1 | tail -1 >> $OUT
cat $OUT
