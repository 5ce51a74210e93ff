#!/bin/bash
# LayerNorm partial reductions of all layers in one launch (single-call step): parity, then same-box A/B against the HEAD build
mkdir -p gpurun_out
timeout 500 python -m pytest tests/test_model_gpu.py -q -x -k "step_graph or known_zero or pinned or driver_epoch or three_optimizer" 2>&1 | grep -v "Warning\|^  warn\|^$" | tail -5 > gpurun_out/lnred.txt
REPS=3 bash scripts/gpu_ab.sh head >> gpurun_out/lnred.txt 2>&1
cat gpurun_out/lnred.txt
