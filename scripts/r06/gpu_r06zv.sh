#!/bin/bash
# 128 x 64 ping-pong tile as the DEFAULT of the narrow launches, DMA pieces spread over COMP, riders on its idle CUs: parity + same-box A/B
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
OUT=$R/gpurun_out/r06_pn_default_ab.txt
bash scripts/box_log.sh > /dev/null 2>&1
GB=$R/tools/bin/gemm_bench; SB=$R/tools/bin/step_bench
{
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py tests/test_xlnet_gpu.py -q -x -k "gemm_nt or gemm_nn or riders or riding" 2>&1 | tail -5
for t in 64 12872; do
  echo "== gemm_bench MB_GEMM_TILE_N768=$t"; MB_GEMM_TILE_N768=$t timeout 120 $GB --T 2400 --nset 24 2>&1 | grep -v "^wgrad\|probe"
done
for rep in 1 2 3; do
  for cfg in "MB_GEMM_TILE_N768=64" "MB_GEMM_TILE_N768=12872" "MB_GEMM_TILE_N768=12872 MB_ADAMW_RIDE_DGRAD=0" "MB_GEMM_TILE_N768=12872 MB_ADAMW_RIDE_PN_PARAMS=262144" "MB_GEMM_TILE_N768=12872 MB_ADAMW_RIDE_PN_PARAMS=655360"; do
    echo "== step B=48 L=50 $cfg"; env $cfg timeout 60 $SB --graph 1 --h2d 2 --steps 100 --warmup 10 2>&1 | grep -o "[0-9.]* ms/step (events)"
  done
done
for rep in 1 2; do
  for cfg in "MB_GEMM_TILE_N768=64" "MB_GEMM_TILE_N768=12872"; do
    echo "== step xlnet B=48 L=50 $cfg"; env $cfg timeout 60 $SB --model xlnet --graph 1 --h2d 2 --steps 60 --warmup 10 2>&1 | grep -o "[0-9.]* ms/step (events)"
    echo "== step C5 B=32 L=128 $cfg"; env $cfg timeout 60 $SB --graph 1 --h2d 2 --batch 32 --seq 128 --visual 35 --steps 60 --warmup 10 2>&1 | grep -o "[0-9.]* ms/step (events)"
  done
done
MB_GEMM_LOG=1 timeout 60 $SB --graph 1 --h2d 2 --steps 3 --warmup 1 2>&1 | grep -E "magbert ride|magbert adamw" | sort | uniq -c | sort -rn | head
} > $OUT 2>&1
cat $OUT
