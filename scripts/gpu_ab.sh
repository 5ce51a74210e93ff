#!/bin/bash
# same-box A/B of builds of the library: arguments = directories under gpurun_ab/ holding another libmagbert_hip.so
# ("current" = the in-tree build).  Alternates the variants REPS times; every run is its own process under a timeout.
mkdir -p gpurun_out
out=gpurun_out/ab.txt
: > $out
REPS=${REPS:-2}
ARGS=${ARGS:---steps 200 --warmup 30 --graph 1 --h2d 2}
for rep in $(seq $REPS); do
  for v in current "$@"; do
    echo "== $v" >> $out
    if [ "$v" = current ]; then timeout 120 tools/bin/step_bench $ARGS >> $out 2>&1
    else LD_LIBRARY_PATH=$PWD/gpurun_ab/$v:$LD_LIBRARY_PATH timeout 120 tools/bin/step_bench $ARGS >> $out 2>&1; fi
  done
done
grep -B1 "ms/step" $out | grep -v "^--" | paste - - | awk '{print $2, $12, $13}'
