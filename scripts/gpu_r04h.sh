#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r04h; mkdir -p $O; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_dp_gpu.py -x -q -k "sharded or rccl" > $O/test_shard.txt 2>&1
tail -n 30 $O/test_shard.txt | cut -c1-250
