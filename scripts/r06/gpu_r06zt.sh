#!/bin/bash
# The 128 x 64 eight-wave ping-pong tile (tile code 12872, MB_GEMM_TILE_N768=12872) for the N = 768 launches: parity, stand-alone, in the step
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
OUT=$R/gpurun_out/r06_pn_first_ab.txt
bash scripts/box_log.sh > /dev/null 2>&1
{
timeout 600 python -m pytest tests/test_ops_gpu.py -q -x -k "gemm_nt or gemm_nn" 2>&1 | tail -5
GB=$R/tools/bin/gemm_bench; SB=$R/tools/bin/step_bench
for rep in 1 2; do
  for t in 64 12872; do
    echo "== gemm_bench MB_GEMM_TILE_N768=$t (rep $rep)"; MB_GEMM_TILE_N768=$t timeout 120 $GB --T 2400 --nset 24 2>&1 | grep -v "^wgrad\|probe"
  done
done
for rep in 1 2 3; do
  for t in 64 12872; do
    echo "== step B=48 L=50 MB_GEMM_TILE_N768=$t"; MB_GEMM_TILE_N768=$t timeout 60 $SB --graph 1 --h2d 2 --steps 100 --warmup 10 2>&1 | grep -o "[0-9.]* ms/step (events)"
  done
done
for rep in 1 2; do
  for t in 64 12872; do
    echo "== step xlnet B=48 L=50 MB_GEMM_TILE_N768=$t"; MB_GEMM_TILE_N768=$t timeout 60 $SB --model xlnet --graph 1 --h2d 2 --steps 60 --warmup 10 2>&1 | grep -o "[0-9.]* ms/step (events)"
  done
done
} > $OUT 2>&1
cat $OUT
