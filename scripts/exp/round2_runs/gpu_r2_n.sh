#!/bin/bash
# where the gelu' epilogue and the attention backward spend their time
mkdir -p gpurun_out
out=gpurun_out/phases2.txt
: > $out
run() { echo "== $*" >> $out; env "$@" timeout 60 tools/bin/gemm_bench --T 2400 --reps 96 --only "dgrad ffn2" "${EXTRA[@]}" >> $out 2>&1; }
EXTRA=()
run MB_GEMM_DBG=0
run MB_GEMM_DBG=16
EXTRA=(--trace 1)
run MB_GEMM_TRACE=1 MB_GEMM_DBG=16
echo "== attention" >> $out
timeout 60 tools/bin/attn_bench >> $out 2>&1
MB_ATTN_TRACE=1 timeout 60 tools/bin/attn_bench >> $out 2>&1
MB_ATTN_TRACE=1 timeout 60 tools/bin/attn_bench --p 0 >> $out 2>&1
MB_ATTN_TRACE=1 timeout 60 tools/bin/attn_bench --batch 32 --seq 128 >> $out 2>&1
cat $out
