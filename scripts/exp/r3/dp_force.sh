#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out; out=gpurun_out/r3_dp_force.txt; : > $out
echo "== single-call step" >> $out; timeout 200 python bench.py --cpu-baseline 0 --roofline 0 --steps 40 --warmup 8 2>&1 | grep -a "^{\|Error\|error\|Traceback" | tail -3 | cut -c1-1500 >> $out
echo "== MB_DP_FORCE=1 (1-rank RCCL group, fp32 wire)" >> $out; MB_DP_FORCE=1 timeout 200 python bench.py --cpu-baseline 0 --roofline 0 --steps 40 --warmup 8 2>&1 | grep -a "^{\|Error\|error\|Traceback" | tail -3 | cut -c1-1500 >> $out
echo "== MB_DP_FORCE=1 MB_DP_GRAD_DTYPE=bf16" >> $out; MB_DP_FORCE=1 MB_DP_GRAD_DTYPE=bf16 timeout 200 python bench.py --cpu-baseline 0 --roofline 0 --steps 40 --warmup 8 2>&1 | grep -a "^{\|Error\|error\|Traceback" | tail -3 | cut -c1-1500 >> $out
cat $out
