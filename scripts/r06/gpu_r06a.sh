#!/bin/bash
# round 6, first visit: parity of the ping-pong tile + stand-alone A/B of the grouped weight gradient and the N >= 2304 GEMMs
mkdir -p gpurun_out/r06a
O=gpurun_out/r06a
(timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider -k "gemm" 2>&1 | tail -30) > $O/pytest_gemm.log 2>&1
tail -3 $O/pytest_gemm.log
for rep in 1 2; do
  for wt in 128 256; do
    echo "== wtile $wt (rep $rep)" >> $O/gemm_ab.txt
    timeout 120 tools/bin/gemm_bench --only wgrad --wtile $wt --nset 24 >> $O/gemm_ab.txt 2>&1
  done
  for tl in 0 256; do
    echo "== tile $tl (rep $rep)" >> $O/gemm_ab.txt
    timeout 120 tools/bin/gemm_bench --tile $tl --nset 24 --only "f" >> $O/gemm_ab.txt 2>&1
  done
done
for T in 4096; do
  for wt in 128 256; do echo "== T $T wtile $wt" >> $O/gemm_ab.txt; timeout 120 tools/bin/gemm_bench --T $T --only wgrad --wtile $wt --nset 24 >> $O/gemm_ab.txt 2>&1; done
  for tl in 0 256; do echo "== T $T tile $tl" >> $O/gemm_ab.txt; timeout 120 tools/bin/gemm_bench --T $T --tile $tl --nset 24 --only "f" >> $O/gemm_ab.txt 2>&1; done
done
echo "== phases wtile 256" >> $O/gemm_ab.txt
MB_GEMM_TRACE=1 timeout 120 tools/bin/gemm_bench --only wgrad --wtile 256 --nset 24 --trace 1 >> $O/gemm_ab.txt 2>&1
echo "== looptrace wtile 256" >> $O/gemm_ab.txt
MB_GEMM_TRACE=1 LD_LIBRARY_PATH=$PWD/gpurun_ab/looptrace:$LD_LIBRARY_PATH timeout 120 tools/bin/gemm_bench --only wgrad --wtile 256 --nset 24 --looptrace 2 >> $O/gemm_ab.txt 2>&1
echo "== looptrace tile 256 fwd ffn1" >> $O/gemm_ab.txt
MB_GEMM_TRACE=1 LD_LIBRARY_PATH=$PWD/gpurun_ab/looptrace:$LD_LIBRARY_PATH timeout 120 tools/bin/gemm_bench --only "fwd ffn1" --tile 256 --nset 24 --looptrace 2 >> $O/gemm_ab.txt 2>&1
cat $O/gemm_ab.txt
