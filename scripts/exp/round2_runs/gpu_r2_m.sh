#!/bin/bash
# epilogue prefetch + packed GELU: parity of the GEMM launches, phase stamps, timing
mkdir -p gpurun_out
out=gpurun_out/epi_prefetch.txt
: > $out
timeout 300 python -m pytest tests/test_ops_gpu.py -q -k "gemm" 2>&1 | tail -15 >> $out
run() { echo "== $*" >> $out; env "$@" timeout 60 tools/bin/gemm_bench --T 2400 --reps 96 "${EXTRA[@]}" >> $out 2>&1; }
EXTRA=()
run MB_GEMM_TILE_BIG=0
run MB_GEMM_TILE_BIG=0
EXTRA=(--trace 1)
run MB_GEMM_TRACE=1 MB_GEMM_TILE_BIG=0
for v in 0 0; do echo "== step_bench MB_GEMM_TILE_BIG=$v" >> $out; MB_GEMM_TILE_BIG=$v timeout 120 tools/bin/step_bench --steps 200 --warmup 30 --graph 1 --h2d 2 >> $out 2>&1; done
cat $out
