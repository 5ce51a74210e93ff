#!/bin/bash
# ring depth of the non-grouped GEMM kernels, per problem (tools/gemm_bench over rotating operand sets): default vs MB_GEMM_STAGES=13 / 14
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/r04p; O=gpurun_out/r04p
for rep in 1 2; do
for v in 0 13 14; do
  echo "== MB_GEMM_STAGES=$v (0 = default selection)" >> $O/gemm_stages.txt
  if [ $v = 0 ]; then timeout 100 tools/bin/gemm_bench --nset 24 >> $O/gemm_stages.txt 2>&1; else MB_GEMM_STAGES=$v timeout 100 tools/bin/gemm_bench --nset 24 >> $O/gemm_stages.txt 2>&1; fi
done
done
cat $O/gemm_stages.txt | cut -c1-110
