#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
export TMPDIR=/tmp
bash scripts/box_log.sh > /dev/null 2>&1
timeout 1200 python -m pytest tests/ -q -m gpu -x 2>&1 | tail -3 > $R/gpurun_out/r06_gpu_suite_final3.txt
cat $R/gpurun_out/r06_gpu_suite_final3.txt
J='^{"metric'
(timeout 500 python bench.py 2>&1 | grep "$J") > $R/gpurun_out/r06_bench_line_final3.json
(timeout 300 python bench.py --model xlnet --cpu-steps 2 --steps 30 --warmup 6 2>&1 | grep "$J") > $R/gpurun_out/r06_bench_line_xlnet_final3.json
cut -c1-230 $R/gpurun_out/r06_bench_line_final3.json; cut -c1-230 $R/gpurun_out/r06_bench_line_xlnet_final3.json
