#!/bin/bash
# Loop trace + phase picture of the 128 x 64 ping-pong tile on the three long narrow launches
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
OUT=$R/gpurun_out/r06_pn_looptrace.txt
GB=$R/tools/bin/gemm_bench
{
for only in "fwd ffn2" "dgrad ffn1" "dgrad qkv" "fwd out"; do
  for nset in 6 24; do
    echo "== $only, nset $nset, MB_GEMM_TILE_N768=12872 (looptrace build)"
    MB_GEMM_TILE_N768=12872 MB_GEMM_TRACE=1 LD_LIBRARY_PATH=$R/gpurun_ab/lt:$LD_LIBRARY_PATH timeout 120 $GB --only "$only" --nset $nset --looptrace 2 2>&1
  done
done
echo "== 64 x 64, phase picture"
for only in "fwd ffn2" "dgrad ffn1"; do
  MB_GEMM_TRACE=1 timeout 120 $GB --only "$only" --nset 24 --trace 1 2>&1
done
} > $OUT 2>&1
cat $OUT
