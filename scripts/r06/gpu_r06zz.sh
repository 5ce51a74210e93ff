#!/bin/bash
# ping-pong kernels with ONE barrier per k-stage vs two: parity, loop trace, stand-alone, same-box step A/B (all three workloads)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
OUT=$R/gpurun_out/r06_pp_one_barrier.txt
GB=$R/tools/bin/gemm_bench; SB=$R/tools/bin/step_bench
TWO="LD_LIBRARY_PATH=$R/gpurun_ab/twob:$LD_LIBRARY_PATH"
{
timeout 300 python -m pytest tests/test_ops_gpu.py -q -x -k "gemm" 2>&1 | tail -3
for only in "fwd ffn2" "dgrad ffn1"; do
    echo "== $only, nset 6 (looptrace build, one barrier)"
    MB_GEMM_TRACE=1 LD_LIBRARY_PATH=$R/gpurun_ab/lt:$LD_LIBRARY_PATH timeout 120 $GB --only "$only" --nset 6 --looptrace 2 2>&1
done
echo "== wgrad 256, nset 6 (looptrace build, one barrier)"
MB_GEMM_TRACE=1 LD_LIBRARY_PATH=$R/gpurun_ab/lt:$LD_LIBRARY_PATH timeout 120 $GB --only wgrad --wtile 256 --nset 6 --looptrace 2 2>&1
for rep in 1 2; do
  echo "== gemm_bench one barrier"; timeout 120 $GB --T 2400 --nset 24 --wtile 256 2>&1 | grep -v "probe"
  echo "== gemm_bench two barriers"; env $TWO timeout 120 $GB --T 2400 --nset 24 --wtile 256 2>&1 | grep -v "probe"
done
for rep in 1 2 3; do
  echo "== step B=48 L=50 MB_GEMM_TILE_N768=64 two barriers"; env $TWO MB_GEMM_TILE_N768=64 timeout 60 $SB --graph 1 --h2d 2 --steps 100 --warmup 10 2>&1 | grep -o "[0-9.]* ms/step (events)"
  echo "== step B=48 L=50 two barriers"; env $TWO MB_ADAMW_RIDE_PN_PARAMS=262144 timeout 60 $SB --graph 1 --h2d 2 --steps 100 --warmup 10 2>&1 | grep -o "[0-9.]* ms/step (events)"
  echo "== step B=48 L=50 one barrier"; MB_ADAMW_RIDE_PN_PARAMS=262144 timeout 60 $SB --graph 1 --h2d 2 --steps 100 --warmup 10 2>&1 | grep -o "[0-9.]* ms/step (events)"
done
for rep in 1 2; do
  echo "== step xlnet two barriers"; env $TWO MB_ADAMW_RIDE_PN_PARAMS=262144 timeout 60 $SB --model xlnet --graph 1 --h2d 2 --steps 60 --warmup 10 2>&1 | grep -o "[0-9.]* ms/step (events)"
  echo "== step xlnet one barrier"; MB_ADAMW_RIDE_PN_PARAMS=262144 timeout 60 $SB --model xlnet --graph 1 --h2d 2 --steps 60 --warmup 10 2>&1 | grep -o "[0-9.]* ms/step (events)"
  echo "== step C5 two barriers"; env $TWO timeout 60 $SB --graph 1 --h2d 2 --batch 32 --seq 128 --visual 35 --steps 60 --warmup 10 2>&1 | grep -o "[0-9.]* ms/step (events)"
  echo "== step C5 one barrier"; timeout 60 $SB --graph 1 --h2d 2 --batch 32 --seq 128 --visual 35 --steps 60 --warmup 10 2>&1 | grep -o "[0-9.]* ms/step (events)"
done
} > $OUT 2>&1
cat $OUT
