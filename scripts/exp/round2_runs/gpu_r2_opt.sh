#!/bin/bash
# head_mask / inputs_embeds checks
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_model_gpu.py -q -k "head_mask" -s 2>&1 | grep -v "^E    +\|^E        +" | tail -80 > gpurun_out/opt_tests.log
tail -60 gpurun_out/opt_tests.log
