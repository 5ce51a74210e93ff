#!/bin/bash
# round 6: ping-pong v2 (DMA + address arithmetic inside COMP, LOAD = reads only)
mkdir -p gpurun_out/r06g
O=gpurun_out/r06g/pp2.txt
: > $O
(timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider -k "gemm" 2>&1 | tail -5) >> $O 2>&1
for rep in 1 2; do
for nset in 6 24; do
  echo "== nset $nset 128x128" >> $O
  timeout 120 tools/bin/gemm_bench --only wgrad --wtile 128 --nset $nset 2>&1 | grep wgrad >> $O
  echo "== nset $nset 256x128 ping-pong" >> $O
  timeout 120 tools/bin/gemm_bench --only wgrad --wtile 256 --nset $nset 2>&1 | grep wgrad >> $O
done
done
echo "== tile 0 vs 256, nset 24" >> $O
timeout 120 tools/bin/gemm_bench --tile 0 --nset 24 --only "f" >> $O 2>&1
timeout 120 tools/bin/gemm_bench --tile 256 --nset 24 --only "f" >> $O 2>&1
echo "== phases ping-pong nset 24" >> $O
MB_GEMM_TRACE=1 timeout 120 tools/bin/gemm_bench --only wgrad --wtile 256 --nset 24 --trace 1 >> $O 2>&1
for dbg in 0 1 2 4 5 6 7; do
  echo "== ablate MB_GEMM_DBG=$dbg" >> $O
  MB_GEMM_DBG=$dbg MB_GEMM_TRACE=1 LD_LIBRARY_PATH=$PWD/gpurun_ab/ablate:$LD_LIBRARY_PATH timeout 120 tools/bin/gemm_bench --only wgrad --wtile 256 --nset 24 --trace 1 2>&1 | grep "wgrad\|k loop\|stage 0" >> $O
done
echo "== looptrace" >> $O
MB_GEMM_TRACE=1 LD_LIBRARY_PATH=$PWD/gpurun_ab/lt_ablate:$LD_LIBRARY_PATH timeout 120 tools/bin/gemm_bench --only wgrad --wtile 256 --nset 24 --looptrace 2 >> $O 2>&1
cat $O
