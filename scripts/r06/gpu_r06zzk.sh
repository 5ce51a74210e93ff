#!/bin/bash
# the 256 x 64 ping-pong tile for the N = 768 launches at T = 4096 (MOSEI shape): parity, stand-alone, same-box step A/B (MB_GEMM_PT=0 = 64 x 64 as before)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
OUT=$R/gpurun_out/r06_pt_first_ab.txt
GB=$R/tools/bin/gemm_bench; SB=$R/tools/bin/step_bench
export TMPDIR=/tmp
bash scripts/box_log.sh > /dev/null 2>&1
{
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -q -x -k "gemm_nt or gemm_nn or benchmark_like" 2>&1 | tail -3
for rep in 1 2; do
  for pt in 0 1; do
    echo "== gemm_bench T=4096 MB_GEMM_PT=$pt"; MB_GEMM_PT=$pt timeout 120 $GB --T 4096 --nset 12 2>&1 | grep -v "^wgrad\|probe"
  done
done
for rep in 1 2 3; do
  for cfg in "MB_GEMM_PT=0" "MB_GEMM_PT=1" "MB_GEMM_PT=1 MB_ADAMW_RIDE_DGRAD=1"; do
    echo "== step C5 $cfg"; env $cfg timeout 60 $SB --graph 1 --h2d 2 --batch 32 --seq 128 --visual 35 --steps 60 --warmup 10 2>&1 | grep -o "[0-9.]* ms/step (events)"
  done
done
MB_GEMM_LOG=1 timeout 60 $SB --graph 1 --h2d 2 --batch 32 --seq 128 --visual 35 --steps 3 --warmup 1 2>&1 | grep -E "magbert ride|magbert adamw" | sort | uniq -c | sort -rn | head
} > $OUT 2>&1
cat $OUT
