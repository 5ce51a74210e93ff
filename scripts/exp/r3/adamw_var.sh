#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out; out=gpurun_out/r3_adamw_var.txt; : > $out
B=tools/bin/adamw_bench
for rep in 1 2; do
for v in 0 1 2 3 4 5; do echo "== VAR=$v" >> $out; MB_ADAMW_VAR=$v timeout 60 $B >> $out 2>&1; done
done
echo "== VAR=0 zero=0" >> $out; timeout 60 $B --zero 0 >> $out 2>&1
echo "== VAR=1 zero=0" >> $out; MB_ADAMW_VAR=1 timeout 60 $B --zero 0 >> $out 2>&1
for gsz in 1024 2048 8192; do echo "== VAR=1 GRID=$gsz" >> $out; MB_ADAMW_VAR=1 MB_ADAMW_GRID=$gsz timeout 60 $B >> $out 2>&1; done
echo "== NT=0" >> $out; MB_ADAMW_NT=0 timeout 60 $B >> $out 2>&1
cat $out
