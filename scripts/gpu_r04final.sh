#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r04final; mkdir -p $O; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests -m gpu -q > $O/test_all_1.txt 2>&1
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > $O/smoke.txt 2>&1
bash scripts/gpu_artifacts.sh r04 > $O/artifacts_stdout.txt 2>&1
tail -n 30 $O/artifacts_stdout.txt | cut -c1-250
