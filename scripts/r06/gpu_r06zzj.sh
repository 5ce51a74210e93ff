#!/bin/bash
# what is the narrow tile's COMP phase made of?  ablations (MB_GEMM_DBG: 1 no DMA, 2 no MFMA, 4 no fragment reads) under the loop trace
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
OUT=$R/gpurun_out/r06_pn_ablate.txt
GB=$R/tools/bin/gemm_bench
{
for dbg in 0 1 4 5 2 6; do
    echo "== fwd ffn2, nset 6, ablate + looptrace build, MB_GEMM_DBG=$dbg"
    MB_GEMM_DBG=$dbg MB_GEMM_TRACE=1 LD_LIBRARY_PATH=$R/gpurun_ab/lt_ablate:$LD_LIBRARY_PATH timeout 120 $GB --only "fwd ffn2" --nset 6 --looptrace 2 2>&1 | grep -v "^per-layer"
done
} > $OUT 2>&1
cat $OUT
