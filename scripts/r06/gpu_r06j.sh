#!/bin/bash
mkdir -p gpurun_out/r06j
O=gpurun_out/r06j/pp4.txt
: > $O
for rep in 1 2; do
for nset in 6 24; do
  echo "== nset $nset 128x128" >> $O
  timeout 120 tools/bin/gemm_bench --only wgrad --wtile 128 --nset $nset 2>&1 | grep wgrad >> $O
  for dbg in 0 16; do
  echo "== nset $nset 256x128 ping-pong MB_GEMM_DBG=$dbg" >> $O
  MB_GEMM_DBG=$dbg timeout 120 tools/bin/gemm_bench --only wgrad --wtile 256 --nset $nset 2>&1 | grep wgrad >> $O
  done
done
done
for dbg in 0 16; do
echo "== looptrace nset 6 MB_GEMM_DBG=$dbg" >> $O
MB_GEMM_DBG=$dbg MB_GEMM_TRACE=1 LD_LIBRARY_PATH=$PWD/gpurun_ab/lt:$LD_LIBRARY_PATH timeout 120 tools/bin/gemm_bench --only wgrad --wtile 256 --nset 6 --looptrace 2 >> $O 2>&1
done
cat $O
