#!/bin/bash
# bench line with the new kernels + re-tests of the big tiles now that the ping-pong loop has one barrier
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
OUT=$R/gpurun_out/r06_onebarrier_retests.txt
SB=$R/tools/bin/step_bench
bash scripts/box_log.sh > /dev/null 2>&1
timeout 600 python bench.py > $R/gpurun_out/r06_bench_line_pn.json 2> $R/gpurun_out/r06_bench_line_pn.err
{
for rep in 1 2; do
  for cfg in "MB_X=0" "MB_GEMM_TILE_BIG=1"; do
    echo "== step B=48 L=50 $cfg"; env $cfg timeout 60 $SB --graph 1 --h2d 2 --steps 100 --warmup 10 2>&1 | grep -o "[0-9.]* ms/step (events)"
  done
  for cfg in "MB_X=0" "MB_GROUP_WGRAD=256" "MB_GEMM_TILE_BIG=1"; do
    echo "== step xlnet $cfg"; env $cfg timeout 60 $SB --model xlnet --graph 1 --h2d 2 --steps 60 --warmup 10 2>&1 | grep -o "[0-9.]* ms/step (events)"
  done
  for cfg in "MB_X=0" "MB_GEMM_PN_MAX=384" "MB_GEMM_TILE_BIG=2"; do
    echo "== step C5 $cfg"; env $cfg timeout 60 $SB --graph 1 --h2d 2 --batch 32 --seq 128 --visual 35 --steps 60 --warmup 10 2>&1 | grep -o "[0-9.]* ms/step (events)"
  done
done
} > $OUT 2>&1
cat $OUT; tail -c 600 $R/gpurun_out/r06_bench_line_pn.err
