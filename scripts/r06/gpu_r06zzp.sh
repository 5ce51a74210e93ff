#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_xlnet_gpu.py -q 2>&1 | tail -3
timeout 300 python bench.py --model xlnet --cpu-baseline 0 --steps 30 --warmup 6 2>/dev/null | cut -c1-200
