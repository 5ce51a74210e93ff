#!/bin/bash
# round 6: ping-pong v2 + immediate-offset DMA + whole-row fp32 stores (MB_GEMM_DBG=32: the old eight-column store form)
mkdir -p gpurun_out/r06i
O=gpurun_out/r06i/pp3.txt
: > $O
(timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider -k "gemm" 2>&1 | tail -5) >> $O 2>&1
for rep in 1 2; do
for nset in 6 24; do
  for dbg in 0 32; do
  echo "== nset $nset 128x128 MB_GEMM_DBG=$dbg" >> $O
  MB_GEMM_DBG=$dbg timeout 120 tools/bin/gemm_bench --only wgrad --wtile 128 --nset $nset 2>&1 | grep wgrad >> $O
  echo "== nset $nset 256x128 ping-pong MB_GEMM_DBG=$dbg" >> $O
  MB_GEMM_DBG=$dbg timeout 120 tools/bin/gemm_bench --only wgrad --wtile 256 --nset $nset 2>&1 | grep wgrad >> $O
  done
done
done
echo "== phases ping-pong nset 24" >> $O
MB_GEMM_TRACE=1 timeout 120 tools/bin/gemm_bench --only wgrad --wtile 256 --nset 24 --trace 1 >> $O 2>&1
echo "== phases 128 nset 24" >> $O
MB_GEMM_TRACE=1 timeout 120 tools/bin/gemm_bench --only wgrad --wtile 128 --nset 24 --trace 1 >> $O 2>&1
echo "== looptrace nset 6" >> $O
MB_GEMM_TRACE=1 LD_LIBRARY_PATH=$PWD/gpurun_ab/lt:$LD_LIBRARY_PATH timeout 120 tools/bin/gemm_bench --only wgrad --wtile 256 --nset 6 --looptrace 2 >> $O 2>&1
cat $O
