#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out; out=gpurun_out/r3_deep.txt; : > $out
echo "== ops tests with the deep kernel" >> $out
MB_GEMM_TILE_N768=12864 MB_GEMM_DEEP=3 timeout 300 python -m pytest tests/test_ops_gpu.py -x -q -m gpu 2>&1 | tail -4 >> $out
for rep in 1 2; do
echo "== default" >> $out; timeout 60 tools/bin/gemm_bench >> $out 2>&1
echo "== MB_GEMM_TILE_N768=12864 MB_GEMM_DEEP=3" >> $out; MB_GEMM_TILE_N768=12864 MB_GEMM_DEEP=3 timeout 60 tools/bin/gemm_bench >> $out 2>&1
done
for rep in 1 2; do
echo "== step default" >> $out; timeout 120 tools/bin/step_bench --steps 200 --warmup 30 --graph 1 --h2d 2 >> $out 2>&1
echo "== step deep" >> $out; MB_GEMM_TILE_N768=12864 MB_GEMM_DEEP=3 timeout 120 tools/bin/step_bench --steps 200 --warmup 30 --graph 1 --h2d 2 >> $out 2>&1
done
cat $out
