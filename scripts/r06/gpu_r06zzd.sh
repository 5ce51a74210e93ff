#!/bin/bash
# full GPU suite on the tree with the 128 x 64 ping-pong tile + one-barrier loops as defaults
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/ -q -m gpu -x 2>&1 | tail -15 > $R/gpurun_out/r06_gpu_suite_pn.txt
cat $R/gpurun_out/r06_gpu_suite_pn.txt
