#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_model_gpu.py tests/test_xlnet_gpu.py -x -q -m gpu -k "position_ids or inputs_embeds or optional_outputs or base_model or head_mask" 2>&1 | tail -30 > gpurun_out/r3_new_tests.txt
cat gpurun_out/r3_new_tests.txt
