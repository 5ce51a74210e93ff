#!/bin/bash
# round 6: attention backward at L = 128 -- round-5 kernel (222 registers, one workgroup per CU) | restructured (148, still one per CU) | capped at 128 (two per CU)
mkdir -p gpurun_out/r06u
O=gpurun_out/r06u/attn_bwd128.txt
: > $O
(timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider -k "attention" 2>&1 | tail -3) >> $O 2>&1
(MB_LIB_DIR=$PWD/gpurun_ab/attn_occ4 timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider -k "attention" 2>&1 | tail -3) >> $O 2>&1
for rep in 1 2; do
for v in attn_old current attn_occ4; do
  for shape in "--batch 32 --seq 128" "--batch 48 --seq 50"; do
    echo "== $v $shape" >> $O
    if [ "$v" = current ]; then timeout 60 tools/bin/attn_bench $shape --reps 200 2>&1 | tail -3 >> $O
    else LD_LIBRARY_PATH=$PWD/gpurun_ab/$v:$LD_LIBRARY_PATH timeout 60 tools/bin/attn_bench $shape --reps 200 2>&1 | tail -3 >> $O; fi
  done
done
done
ARGS="--steps 200 --warmup 30 --graph 1 --h2d 2 --batch 32 --seq 128 --visual 35"
for rep in 1 2; do
for v in attn_old current attn_occ4; do
  echo "== step C5 $v" >> $O
  if [ "$v" = current ]; then timeout 120 tools/bin/step_bench $ARGS 2>&1 | grep "ms/step" | cut -c64-110 >> $O
  else LD_LIBRARY_PATH=$PWD/gpurun_ab/$v:$LD_LIBRARY_PATH timeout 120 tools/bin/step_bench $ARGS 2>&1 | grep "ms/step" | cut -c64-110 >> $O; fi
done
done
cat $O
