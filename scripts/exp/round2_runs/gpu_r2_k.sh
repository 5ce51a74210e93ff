#!/bin/bash
# 256 x 128 eight-wave tiles: parity of the GEMM launches, then A/B in gemm_bench and step_bench
mkdir -p gpurun_out
out=gpurun_out/big_tile.txt
: > $out
timeout 300 python -m pytest tests/test_ops_gpu.py -q -k "gemm" 2>&1 | tail -15 >> $out
run() { echo "== $*" >> $out; env "$@" timeout 60 tools/bin/gemm_bench --T 2400 --reps 96 >> $out 2>&1; }
run MB_GEMM_TILE_BIG=0
run MB_GEMM_TILE_BIG=1
run MB_GEMM_TILE_BIG=1 MB_GEMM_DBG=1
run MB_GEMM_TILE_BIG=1 MB_GEMM_DBG=6
for v in 0 1 0 1; do echo "== step_bench MB_GEMM_TILE_BIG=$v" >> $out; MB_GEMM_TILE_BIG=$v timeout 120 tools/bin/step_bench --steps 200 --warmup 30 --graph 1 --h2d 2 >> $out 2>&1; done
cat $out
