#!/bin/bash
# 64-byte k rows (MB_GEMM_STAGES=2x: x ring slots of 64-byte rows for every gemm2 launch -- 64 x 64 tiles: 24 / 32 / 40 KB =
# six / five / four blocks per CU; 128 x 128 tiles stay at two, their epilogue tile needs 64 KB) against the default selection, same box, in the step
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
bash scripts/box_log.sh > /dev/null 2>&1
for rep in 1 2 3; do
  for cfg in "MB_ADAMW_RIDE_DGRAD=0" "MB_ADAMW_RIDE_DGRAD=0 MB_GEMM_STAGES=23" "MB_ADAMW_RIDE_DGRAD=0 MB_GEMM_STAGES=24" "MB_ADAMW_RIDE_DGRAD=0 MB_GEMM_STAGES=25" "MB_ADAMW_RIDE_DGRAD=0 MB_GEMM_STAGES=13" "MB_ADAMW_RIDE_DGRAD=0 MB_GEMM_64_STAGES=4"; do
    echo "== $cfg"; env $cfg timeout 60 $R/tools/bin/step_bench --graph 1 --h2d 2 --steps 300 --warmup 20 2>&1 | grep -o "[0-9.]* ms/step (events)"
  done
done
echo "== C5 (B=32 L=128)"
for rep in 1 2; do
  for cfg in "MB_ADAMW_RIDE_DGRAD=0" "MB_ADAMW_RIDE_DGRAD=0 MB_GEMM_STAGES=23" "MB_ADAMW_RIDE_DGRAD=0 MB_GEMM_STAGES=24"; do
    echo "== $cfg"; env $cfg timeout 60 $R/tools/bin/step_bench --graph 1 --h2d 2 --batch 32 --seq 128 --visual 35 --steps 100 --warmup 10 2>&1 | grep -o "[0-9.]* ms/step (events)"
  done
done
echo "== stand-alone, tools/bin/gemm_bench (per-layer GEMM chain, rotating operand sets)"
for cfg in "X=0" "MB_GEMM_STAGES=23" "MB_GEMM_STAGES=24"; do echo "== $cfg"; env $cfg timeout 60 $R/tools/bin/gemm_bench 2>&1 | tail -n 11; done
