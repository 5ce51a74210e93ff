#!/bin/bash
# piggy-backed prefetch, second batch (attention backward touches the GELU output for the weight gradients; MAG-XLNet's LayerNorms
# touch its weights): same-box A/B MB_PREFETCH=0 / 1 for the three workloads, then the GPU tests
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out; export TMPDIR=/tmp
OUT=gpurun_out/r3_prefetch_ab2.txt; : > $OUT
S="tools/bin/step_bench --steps 200 --warmup 30 --graph 1 --h2d 2"
for rep in 1 2; do
  for p in 0 1; do echo "== BERT MB_PREFETCH=$p" >> $OUT; MB_PREFETCH=$p timeout 100 $S | tail -1 >> $OUT; done
done
for p in 0 1; do echo "== C5 MB_PREFETCH=$p" >> $OUT; MB_PREFETCH=$p timeout 100 $S --batch 32 --seq 128 --visual 35 | tail -1 >> $OUT; done
for p in 0 1; do echo "== XLNet MB_PREFETCH=$p" >> $OUT; MB_PREFETCH=$p timeout 200 python bench.py --model xlnet --cpu-baseline 0 --roofline 0 --steps 40 --warmup 8 2>&1 | grep '^{' | cut -c1-160 >> $OUT; done
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 >> $OUT
cat $OUT
