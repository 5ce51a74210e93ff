#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
for i in 1 2; do timeout 60 tools/bin/step_bench --graph 1 --h2d 2 --steps 40 --warmup 8; done
(timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -s 2>&1 | grep -v "Warning\|warn" | tail -150) > gpurun_out/r2h_pytest.log 2>&1
grep -n "passed\|failed\|FAILED\|Error" gpurun_out/r2h_pytest.log | tail -12
(timeout 300 python bench.py --steps 40 --warmup 8 --cpu-baseline 0 2>&1 | tail -3) > gpurun_out/r2h_bench.log 2>&1
tail -1 gpurun_out/r2h_bench.log | cut -c1-700
tail -1 gpurun_out/r2h_bench.log | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print(json.dumps(d.get('roofline_hbm'), indent=0)[:1500])"
