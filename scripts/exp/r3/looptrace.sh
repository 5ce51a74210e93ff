#!/bin/bash
# per-iteration stamps inside the k loop (library built with -DMB_GEMM_LOOPTRACE: python scripts/build_variant.py looptrace -DMB_GEMM_LOOPTRACE)
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out; out=gpurun_out/r3_looptrace.txt; : > $out
export LD_LIBRARY_PATH=$PWD/gpurun_ab/looptrace:$LD_LIBRARY_PATH
MB_GEMM_TRACE=1 timeout 120 tools/bin/gemm_bench --looptrace 1 >> $out 2>&1
cat $out
