#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
SB="timeout 60 $R/tools/bin/step_bench --graph 1 --h2d 2 --steps 40 --warmup 8"
{
for v in 0 1 0 1; do echo "== ADAMW_NT=$v"; MB_ADAMW_NT=$v $SB; done
C5="--batch 32 --seq 128 --visual 35 --steps 24 --warmup 6"
echo "== C5 default (64x64 quarter tiles)"; $SB $C5
echo "== C5 N768 tile 128";   MB_GEMM_TILE_N768=128 $SB $C5
echo "== C5 N768 tile 128x64"; MB_GEMM_TILE_N768=12864 $SB $C5
echo "== C5 default again"; $SB $C5
} 2>&1 | tee gpurun_out/r2g_step_bench.log
