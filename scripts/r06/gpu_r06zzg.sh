#!/bin/bash
# final bench lines (three workloads) + box log
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
export TMPDIR=/tmp
bash scripts/box_log.sh > /dev/null 2>&1
J='^{"metric'
(timeout 500 python bench.py 2>&1 | grep "$J") > $R/gpurun_out/r06_bench_line_final2.json
(timeout 300 python bench.py --dataset mosei --seq 128 --batch 32 --cpu-baseline 0 --steps 30 --warmup 6 2>&1 | grep "$J") > $R/gpurun_out/r06_bench_line_c5_final2.json
(timeout 300 python bench.py --model xlnet --cpu-steps 2 --steps 30 --warmup 6 2>&1 | grep "$J") > $R/gpurun_out/r06_bench_line_xlnet_final2.json
cut -c1-300 $R/gpurun_out/r06_bench_line_final2.json
