#!/bin/bash
# same-box A/B of two builds of the library: $1 = directory under gpurun_ab/ holding the other libmagbert_hip.so
mkdir -p gpurun_out
out=gpurun_out/ab_$1.txt
: > $out
for rep in 1 2 3; do
  echo "== current" >> $out
  timeout 120 tools/bin/step_bench --steps 200 --warmup 30 --graph 1 --h2d 2 >> $out 2>&1
  echo "== $1" >> $out
  LD_LIBRARY_PATH=$PWD/gpurun_ab/$1:$LD_LIBRARY_PATH timeout 120 tools/bin/step_bench --steps 200 --warmup 30 --graph 1 --h2d 2 >> $out 2>&1
done
cat $out
