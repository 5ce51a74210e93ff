#!/bin/bash
# quick check of a GEMM change: operator parity tests, the nine GEMM launches of a layer, whole steps
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out; out=gpurun_out/r3_quick.txt; : > $out
timeout 300 python -m pytest tests/test_ops_gpu.py -x -q -m gpu 2>&1 | tail -5 >> $out
echo "== gemm_bench" >> $out; timeout 60 tools/bin/gemm_bench >> $out 2>&1
echo "== gemm_bench" >> $out; timeout 60 tools/bin/gemm_bench >> $out 2>&1
for i in 1 2; do echo "== step" >> $out; timeout 120 tools/bin/step_bench --steps 200 --warmup 30 --graph 1 --h2d 2 >> $out 2>&1; done
cat $out
