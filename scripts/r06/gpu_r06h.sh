#!/bin/bash
mkdir -p gpurun_out/r06h
O=gpurun_out/r06h/pp2.txt
: > $O
for nset in 6 24; do
for dbg in 0 1 2 4 5 6 3 7; do
  echo "== nset $nset ablate MB_GEMM_DBG=$dbg" >> $O
  MB_GEMM_DBG=$dbg MB_GEMM_TRACE=1 LD_LIBRARY_PATH=$PWD/gpurun_ab/ablate:$LD_LIBRARY_PATH timeout 120 tools/bin/gemm_bench --only wgrad --wtile 256 --nset $nset --trace 1 2>&1 | grep "wgrad\|k loop\|stage 0" >> $O
done
echo "== looptrace nset $nset" >> $O
MB_GEMM_TRACE=1 LD_LIBRARY_PATH=$PWD/gpurun_ab/lt:$LD_LIBRARY_PATH timeout 120 tools/bin/gemm_bench --only wgrad --wtile 256 --nset $nset --looptrace 2 >> $O 2>&1
done
cat $O
