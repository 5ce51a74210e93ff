#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_xlnet_gpu.py -x -q -m gpu 2>&1 | tail -25 > gpurun_out/r3_xl128.txt
cat gpurun_out/r3_xl128.txt
