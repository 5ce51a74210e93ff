#!/bin/bash
# per-block phase stamps of the GEMM launches (128 x 128 / k-split 64 x 64 vs 256 x 128 tiles)
mkdir -p gpurun_out
out=gpurun_out/gemm_phases.txt
: > $out
run() { echo "== $*" >> $out; env "$@" timeout 60 tools/bin/gemm_bench --T 2400 --reps 48 --trace 1 >> $out 2>&1; }
run MB_GEMM_TRACE=1 MB_GEMM_TILE_BIG=0
run MB_GEMM_TRACE=1 MB_GEMM_TILE_BIG=1
cat $out
