#!/bin/bash
# round 6: where a ping-pong k-stage goes -- per-phase stamps under ablation (MB_GEMM_DBG: 1 no DMA, 2 no MFMA, 4 no fragment reads, 16 no s_setprio)
mkdir -p gpurun_out/r06b
O=gpurun_out/r06b/pp_looptrace.txt
: > $O
for dbg in 0 16 1 2 4 6 5; do
  echo "== MB_GEMM_DBG=$dbg" >> $O
  MB_GEMM_DBG=$dbg MB_GEMM_TRACE=1 LD_LIBRARY_PATH=$PWD/gpurun_ab/lt_ablate:$LD_LIBRARY_PATH timeout 120 tools/bin/gemm_bench --only wgrad --wtile 256 --nset 24 --looptrace 2 >> $O 2>&1
done
for dbg in 0 16; do
  echo "== untraced MB_GEMM_DBG=$dbg" >> $O
  MB_GEMM_DBG=$dbg timeout 120 tools/bin/gemm_bench --only wgrad --wtile 256 --nset 24 >> $O 2>&1
done
cat $O
