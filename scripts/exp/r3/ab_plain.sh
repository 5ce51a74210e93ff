#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
REPS=3 bash scripts/gpu_ab.sh plainloop > gpurun_out/r3_ab_plain.txt 2>&1
for v in current plainloop; do
  echo "== gemm_bench $v" >> gpurun_out/r3_ab_plain.txt
  if [ $v = current ]; then timeout 60 tools/bin/gemm_bench >> gpurun_out/r3_ab_plain.txt 2>&1; else LD_LIBRARY_PATH=$PWD/gpurun_ab/$v:$LD_LIBRARY_PATH timeout 60 tools/bin/gemm_bench >> gpurun_out/r3_ab_plain.txt 2>&1; fi
done
ARGS="--graph 1 --h2d 2 --steps 25 --warmup 5" bash scripts/gpu_kstats.sh plainloop > /dev/null 2>&1
cat gpurun_out/r3_ab_plain.txt
