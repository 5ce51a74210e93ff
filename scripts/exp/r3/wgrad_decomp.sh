#!/bin/bash
# round 3, GPU call 1: where does the grouped weight-gradient launch spend its time (ablation switches now reach the grouped kernel)
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out; out=gpurun_out/r3_wgrad_decomp.txt; : > $out
G=tools/bin/gemm_bench
for d in 0 1 2 4 6 0; do echo "== MB_GEMM_DBG=$d" >> $out; MB_GEMM_DBG=$d timeout 60 $G --only wgrad >> $out 2>&1; done
for s in 2 24 25 3 2; do echo "== MB_GROUP_STAGES=$s" >> $out; MB_GROUP_STAGES=$s timeout 60 $G --only wgrad >> $out 2>&1; done
echo "== trace" >> $out; MB_GEMM_TRACE=1 timeout 60 $G --only wgrad --trace 1 >> $out 2>&1
echo "== trace stages 24" >> $out; MB_GROUP_STAGES=24 MB_GEMM_TRACE=1 timeout 60 $G --only wgrad --trace 1 >> $out 2>&1
echo "== all" >> $out; timeout 60 $G >> $out 2>&1
echo "== step" >> $out; timeout 120 tools/bin/step_bench --steps 200 --warmup 30 --graph 1 --h2d 2 >> $out 2>&1
cat $out
