#!/bin/bash
# pn: loop trace with the DMA pieces spread; rider budget scan
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
OUT=$R/gpurun_out/r06_pn_budget.txt
GB=$R/tools/bin/gemm_bench; SB=$R/tools/bin/step_bench
{
for only in "fwd ffn2" "dgrad ffn1"; do
    echo "== $only, nset 6 (looptrace build, DMA pieces spread)"
    MB_GEMM_TRACE=1 LD_LIBRARY_PATH=$R/gpurun_ab/lt:$LD_LIBRARY_PATH timeout 120 $GB --only "$only" --nset 6 --looptrace 2 2>&1
done
for rep in 1 2; do
  for cfg in "MB_ADAMW_RIDE_PN_PARAMS=131072" "MB_ADAMW_RIDE_PN_PARAMS=196608" "MB_ADAMW_RIDE_PN_PARAMS=262144" "MB_ADAMW_RIDE_PN_PARAMS=327680" "MB_ADAMW_RIDE_DGRAD=1 MB_ADAMW_RIDE_PN_PARAMS=262144" ; do
    echo "== step B=48 L=50 $cfg"; env $cfg timeout 60 $SB --graph 1 --h2d 2 --steps 100 --warmup 10 2>&1 | grep -o "[0-9.]* ms/step (events)"
  done
  for cfg in "MB_GEMM_TILE_N768=64" "MB_ADAMW_RIDE_PN_PARAMS=196608" "MB_ADAMW_RIDE_PN_PARAMS=262144" "MB_ADAMW_RIDE_DGRAD=0"; do
    echo "== step xlnet B=48 L=50 $cfg"; env $cfg timeout 60 $SB --model xlnet --graph 1 --h2d 2 --steps 60 --warmup 10 2>&1 | grep -o "[0-9.]* ms/step (events)"
  done
done
} > $OUT 2>&1
cat $OUT
