#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r3_tests.txt
cat gpurun_out/r3_tests.txt
