#!/bin/bash
mkdir -p gpurun_out
out=gpurun_out/n768.txt
: > $out
for v in 64 12864 64 12864; do echo "== MB_GEMM_TILE_N768=$v" >> $out; MB_GEMM_TILE_N768=$v timeout 60 tools/bin/gemm_bench --T 2400 --reps 96 --only "out" >> $out 2>&1; done
cat $out
