#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out; out=gpurun_out/r3_dp_graph.txt; : > $out
timeout 600 python -m pytest tests/test_dp_gpu.py -x -q -m gpu 2>&1 | tail -6 >> $out
B="python bench.py --cpu-baseline 0 --roofline 0 --steps 40 --warmup 8"
for rep in 1 2; do
echo "== single-call step" >> $out; timeout 200 $B 2>&1 | grep -a "^{\|Error\|error\|Traceback" | tail -3 | cut -c1-1500 >> $out
echo "== MB_DP_FORCE=1 stage graphs" >> $out; MB_DP_FORCE=1 timeout 200 $B 2>&1 | grep -a "^{\|Error\|error\|Traceback" | tail -3 | cut -c1-1500 >> $out
echo "== MB_DP_FORCE=1 MB_DP_GRAPH=0 (kernel launches from Python)" >> $out; MB_DP_FORCE=1 MB_DP_GRAPH=0 timeout 200 $B 2>&1 | grep -a "^{\|Error\|error\|Traceback" | tail -3 | cut -c1-1500 >> $out
done
cat $out
