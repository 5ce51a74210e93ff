#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_model_gpu.py tests/test_ops_gpu.py -x -q -m gpu -k "cross_entropy or other_hidden or fused_training or mag_module" 2>&1 | tail -30 > gpurun_out/r3_misc_tests.txt
cat gpurun_out/r3_misc_tests.txt
