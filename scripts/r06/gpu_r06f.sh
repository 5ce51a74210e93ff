#!/bin/bash
# round 6: the 256 x 128 tile as ONE wave per SIMD (4 waves, 128 x 64 wave tiles; the existing pipelined two-slot loop) vs ping-pong vs 128^2
mkdir -p gpurun_out/r06f
O=gpurun_out/r06f/w1.txt
: > $O
(timeout 300 python -m pytest tests/test_ops_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider -k "grouped" 2>&1 | tail -3) >> $O 2>&1
for rep in 1 2; do
for nset in 6 24; do
  echo "== nset $nset 128x128" >> $O
  MB_GEMM_TRACE=0 timeout 120 tools/bin/gemm_bench --only wgrad --wtile 128 --nset $nset 2>&1 | grep wgrad >> $O
  echo "== nset $nset 256x128 one wave per SIMD" >> $O
  MB_GROUP_BIG=1 timeout 120 tools/bin/gemm_bench --only wgrad --wtile 256 --nset $nset 2>&1 | grep wgrad >> $O
  echo "== nset $nset 256x128 ping-pong" >> $O
  MB_GROUP_BIG=2 timeout 120 tools/bin/gemm_bench --only wgrad --wtile 256 --nset $nset 2>&1 | grep wgrad >> $O
done
done
echo "== phases, one wave per SIMD, nset 24" >> $O
MB_GROUP_BIG=1 MB_GEMM_TRACE=1 timeout 120 tools/bin/gemm_bench --only wgrad --wtile 256 --nset 24 --trace 1 >> $O 2>&1
cat $O
