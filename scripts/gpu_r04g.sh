#!/bin/bash
# tile-selection experiments at T = 4096 (the per-GPU shape of BASELINE configs[4]) and T = 2400, stand-alone GEMM launches over
# rotating operand sets (cold: --nset 24 ~ in-step conditions)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r04g; mkdir -p $O; cd $R
GB=$R/tools/bin/gemm_bench
{
for T in 4096 2400; do
for ns in 24; do
  echo "=== T=$T nset=$ns default"; timeout 120 $GB --T $T --nset $ns
  echo "=== T=$T nset=$ns MB_GEMM_TILE_N768=12864"; MB_GEMM_TILE_N768=12864 timeout 120 $GB --T $T --nset $ns
  echo "=== T=$T nset=$ns MB_GEMM_TILE_N768=128"; MB_GEMM_TILE_N768=128 timeout 120 $GB --T $T --nset $ns
  echo "=== T=$T nset=$ns MB_GEMM_TILE_BIG=2"; MB_GEMM_TILE_BIG=2 timeout 120 $GB --T $T --nset $ns
  echo "=== T=$T nset=$ns MB_GEMM_KSPLIT=0"; MB_GEMM_KSPLIT=0 timeout 120 $GB --T $T --nset $ns
  echo "=== T=$T nset=$ns MB_GROUP_WGRAD stages 3"; MB_GROUP_STAGES=3 timeout 120 $GB --T $T --nset $ns --only wgrad
done; done
} > $O/gemm_tiles.txt 2>&1
cat $O/gemm_tiles.txt
