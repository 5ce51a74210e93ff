#!/bin/bash
# round 6: ping-pong tile, untraced ablation + LOAD-phase priority (MB_GEMM_DBG: 1 no DMA, 2 no MFMA, 4 no reads, 16 no COMP prio, 32/64/96 LOAD prio 1/2/3)
mkdir -p gpurun_out/r06c
O=gpurun_out/r06c/pp_ablate.txt
: > $O
for rep in 1 2; do
for dbg in 0 16 32 64 96 112 1 2 4 3 5 6 7; do
  echo "== MB_GEMM_DBG=$dbg" >> $O
  MB_GEMM_DBG=$dbg MB_GEMM_TRACE=1 LD_LIBRARY_PATH=$PWD/gpurun_ab/ablate:$LD_LIBRARY_PATH timeout 120 tools/bin/gemm_bench --only wgrad --wtile 256 --nset 24 --trace 1 2>&1 | grep "wgrad\|k loop\|stage 0" >> $O
done
done
for dbg in 96 112; do
  echo "== traced MB_GEMM_DBG=$dbg" >> $O
  MB_GEMM_DBG=$dbg MB_GEMM_TRACE=1 LD_LIBRARY_PATH=$PWD/gpurun_ab/lt_ablate:$LD_LIBRARY_PATH timeout 120 tools/bin/gemm_bench --only wgrad --wtile 256 --nset 24 --looptrace 2 >> $O 2>&1
done
cat $O
