#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
(timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -s 2>&1 | grep -v "Warning\|warn" | tail -120) > gpurun_out/r2f_pytest.log 2>&1
grep -n "passed\|failed\|FAILED\|Error" gpurun_out/r2f_pytest.log | tail -15
(timeout 300 python bench.py --steps 40 --warmup 8 --cpu-baseline 0 2>&1 | tail -3) > gpurun_out/r2f_bench.log 2>&1
tail -1 gpurun_out/r2f_bench.log | cut -c1-900
