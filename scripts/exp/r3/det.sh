#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out; out=gpurun_out/r3_det.txt; : > $out
timeout 600 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "deterministic" -s 2>&1 | tail -15 >> $out
for v in 0 1 0 1; do echo "== MB_DETERMINISTIC=$v" >> $out; MB_DETERMINISTIC=$v timeout 120 tools/bin/step_bench --steps 200 --warmup 30 --graph 1 --h2d 2 >> $out 2>&1; done
cat $out
