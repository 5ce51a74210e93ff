#!/bin/bash
# what bounds the 128 x 128 tile kernels at T = 2400: debug decomposition + ring depth
mkdir -p gpurun_out
out=gpurun_out/gemm_decomp.txt
: > $out
run() { echo "== $*" >> $out; env "$@" timeout 60 tools/bin/gemm_bench --T 2400 --reps 96 >> $out 2>&1; }
run MB_GEMM_DBG=0
run MB_GEMM_DBG=1
run MB_GEMM_DBG=2
run MB_GEMM_DBG=4
run MB_GEMM_DBG=6
run MB_GEMM_STAGES=13
run MB_GEMM_STAGES=14
run MB_GEMM_DBG=0
cat $out
