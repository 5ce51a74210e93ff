#!/bin/bash
# s_setprio in COMP with the one-barrier loop: MB_GEMM_DBG=16 turns it off
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
OUT=$R/gpurun_out/r06_pp_setprio_onebarrier.txt
SB=$R/tools/bin/step_bench
{
for rep in 1 2 3; do
  for cfg in "MB_X=0" "MB_GEMM_DBG=16"; do
    echo "== step B=48 L=50 $cfg"; env $cfg timeout 60 $SB --graph 1 --h2d 2 --steps 100 --warmup 10 2>&1 | grep -o "[0-9.]* ms/step (events)"
  done
done
} > $OUT 2>&1
cat $OUT
