#!/bin/bash
# pn with k-split waves (KSW) vs 32 x 32 outputs per wave: parity, loop trace, stand-alone, same-box step A/B
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
OUT=$R/gpurun_out/r06_pn_ksw.txt
GB=$R/tools/bin/gemm_bench; SB=$R/tools/bin/step_bench
{
timeout 300 python -m pytest tests/test_ops_gpu.py -q -x -k "gemm_nt or gemm_nn" 2>&1 | tail -3
for only in "fwd ffn2" "dgrad ffn1"; do
    echo "== $only, nset 6 (looptrace build, KSW)"
    MB_GEMM_TRACE=1 LD_LIBRARY_PATH=$R/gpurun_ab/lt:$LD_LIBRARY_PATH timeout 120 $GB --only "$only" --nset 6 --looptrace 2 2>&1
done
for rep in 1 2; do
  echo "== gemm_bench KSW"; timeout 120 $GB --T 2400 --nset 24 2>&1 | grep -v "^wgrad\|probe"
  echo "== gemm_bench 32x32 per wave"; LD_LIBRARY_PATH=$R/gpurun_ab/ksw0:$LD_LIBRARY_PATH timeout 120 $GB --T 2400 --nset 24 2>&1 | grep -v "^wgrad\|probe"
done
for rep in 1 2 3; do
  echo "== step B=48 L=50 MB_GEMM_TILE_N768=64"; MB_GEMM_TILE_N768=64 timeout 60 $SB --graph 1 --h2d 2 --steps 100 --warmup 10 2>&1 | grep -o "[0-9.]* ms/step (events)"
  echo "== step B=48 L=50 KSW PN_PARAMS=262144"; MB_ADAMW_RIDE_PN_PARAMS=262144 timeout 60 $SB --graph 1 --h2d 2 --steps 100 --warmup 10 2>&1 | grep -o "[0-9.]* ms/step (events)"
  echo "== step B=48 L=50 32x32 PN_PARAMS=262144"; MB_ADAMW_RIDE_PN_PARAMS=262144 LD_LIBRARY_PATH=$R/gpurun_ab/ksw0:$LD_LIBRARY_PATH timeout 60 $SB --graph 1 --h2d 2 --steps 100 --warmup 10 2>&1 | grep -o "[0-9.]* ms/step (events)"
done
} > $OUT 2>&1
cat $OUT
