#!/bin/bash
# same-box A/B: build of the previous commit (gpurun_ab/base) vs the in-tree build (transpose-read attention fragments, MAG slabs,
# folded zero fills / colsums, wider ln_reduce, one-round-trip prologue), then per-kernel stats of both and the GPU tests
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$(pwd); OUT=gpurun_out/r3_small_kernels_ab.txt; : > $OUT
S="tools/bin/step_bench --steps 200 --warmup 30 --graph 1 --h2d 2"
for rep in 1 2; do
  echo "== base" >> $OUT; LD_LIBRARY_PATH=$R/gpurun_ab/base timeout 120 $S 2>&1 | tail -1 >> $OUT
  echo "== current" >> $OUT; timeout 120 $S 2>&1 | tail -1 >> $OUT
done
echo "== base C5" >> $OUT; LD_LIBRARY_PATH=$R/gpurun_ab/base timeout 120 $S --batch 32 --seq 128 --visual 35 2>&1 | tail -1 >> $OUT
echo "== current C5" >> $OUT; timeout 120 $S --batch 32 --seq 128 --visual 35 2>&1 | tail -1 >> $OUT
timeout 300 bash scripts/gpu_kstats.sh base > /dev/null 2>&1
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 >> $OUT
cat $OUT
