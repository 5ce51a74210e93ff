#!/bin/bash
# MAG / embedding one-offs (prologue packs the modalities, 128-tile MAG weight gradients, embedding LN slabs, mag_gate_bwd at two
# blocks per CU): same-box step A/B vs gpurun_ab/base, per-kernel stats, GPU tests
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$(pwd); OUT=gpurun_out/r3_oneoffs_ab.txt; : > $OUT
S="tools/bin/step_bench --steps 200 --warmup 30 --graph 1 --h2d 2"
for rep in 1 2; do
  echo "== base" >> $OUT; LD_LIBRARY_PATH=$R/gpurun_ab/base timeout 120 $S 2>&1 | tail -1 >> $OUT
  echo "== current" >> $OUT; timeout 120 $S 2>&1 | tail -1 >> $OUT
  echo "== current MB_MAG_WGRAD_TILE=64" >> $OUT; MB_MAG_WGRAD_TILE=64 timeout 120 $S 2>&1 | tail -1 >> $OUT
  echo "== current MB_PROLOGUE_PACK=0" >> $OUT; MB_PROLOGUE_PACK=0 timeout 120 $S 2>&1 | tail -1 >> $OUT
done
timeout 300 bash scripts/gpu_kstats.sh > /dev/null 2>&1
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 >> $OUT
cat $OUT
