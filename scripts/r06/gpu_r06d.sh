#!/bin/bash
# round 6: is the ping-pong tile's DMA latency-bound (cold operands) or LDS-write-bound?  DMA-only / full at 1, 6, 24 rotating operand sets
mkdir -p gpurun_out/r06d
O=gpurun_out/r06d/pp_dma.txt
: > $O
for nset in 1 6 24; do
for dbg in 6 0 4; do
  echo "== nset $nset MB_GEMM_DBG=$dbg" >> $O
  MB_GEMM_DBG=$dbg MB_GEMM_TRACE=1 LD_LIBRARY_PATH=$PWD/gpurun_ab/ablate:$LD_LIBRARY_PATH timeout 120 tools/bin/gemm_bench --only wgrad --wtile 256 --nset $nset --trace 1 2>&1 | grep "wgrad\|k loop\|stage 0" >> $O
done
  echo "== nset $nset 128x128 tiles" >> $O
  MB_GEMM_TRACE=1 timeout 120 tools/bin/gemm_bench --only wgrad --wtile 128 --nset $nset --trace 1 2>&1 | grep "wgrad\|k loop\|stage 0" >> $O
done
cat $O
