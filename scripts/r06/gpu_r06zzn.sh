#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
export TMPDIR=/tmp
bash scripts/gpu_artifacts.sh r06c > $R/gpurun_out/r06c_artifacts.log 2>&1
timeout 1200 python -m pytest tests/ -q -m gpu 2>&1 | tail -4 > $R/gpurun_out/r06_gpu_suite_final2.txt
cat $R/gpurun_out/r06_gpu_suite_final2.txt; tail -5 $R/gpurun_out/r06c_artifacts.log
