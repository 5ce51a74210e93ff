#!/bin/bash
# GPU tests + GEMM table + whole steps for the in-tree build
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out; out=gpurun_out/r3_full.txt; : > $out
echo "== gemm_bench" >> $out; timeout 60 tools/bin/gemm_bench >> $out 2>&1
for i in 1 2; do echo "== step" >> $out; timeout 120 tools/bin/step_bench --steps 200 --warmup 30 --graph 1 --h2d 2 >> $out 2>&1; done
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 >> $out
cat $out
