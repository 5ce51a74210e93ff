#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out; out=gpurun_out/r3_tile_n768.txt; : > $out
for v in 64 12864 128 64; do echo "== MB_GEMM_TILE_N768=$v" >> $out; MB_GEMM_TILE_N768=$v timeout 60 tools/bin/gemm_bench >> $out 2>&1; done
echo "== KSPLIT=0" >> $out; MB_GEMM_KSPLIT=0 timeout 60 tools/bin/gemm_bench >> $out 2>&1
cat $out
