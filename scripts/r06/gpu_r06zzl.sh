#!/bin/bash
# C5 with the tall tile as default, dgelu second-round riders off: same-box check + attention rider budget at L = 128 + bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
OUT=$R/gpurun_out/r06_pt_defaults.txt
SB=$R/tools/bin/step_bench
export TMPDIR=/tmp
bash scripts/box_log.sh > /dev/null 2>&1
{
for rep in 1 2 3; do
  for cfg in "MB_GEMM_PT=0 MB_ADAMW_RIDE_DGELU_ROUNDS=1" "MB_X=0" "MB_ADAMW_RIDE_DGELU_ROUNDS=1" "MB_ADAMW_RIDE_ATTN_PARAMS=3500000" "MB_ADAMW_RIDE_PN_PARAMS=1200000" "MB_ADAMW_RIDE_PN_PARAMS=2400000"; do
    echo "== step C5 $cfg"; env $cfg timeout 60 $SB --graph 1 --h2d 2 --batch 32 --seq 128 --visual 35 --steps 60 --warmup 10 2>&1 | grep -o "[0-9.]* ms/step (events)"
  done
done
MB_GEMM_LOG=1 timeout 60 $SB --graph 1 --h2d 2 --batch 32 --seq 128 --visual 35 --steps 3 --warmup 1 2>&1 | grep -E "magbert ride|magbert adamw" | sort | uniq -c | sort -rn | head
J='^{"metric'
(timeout 300 python bench.py --dataset mosei --seq 128 --batch 32 --cpu-baseline 0 --steps 30 --warmup 6 2>&1 | grep "$J") > $R/gpurun_out/r06_bench_line_c5_pt.json
cut -c1-200 $R/gpurun_out/r06_bench_line_c5_pt.json
} > $OUT 2>&1
cat $OUT
