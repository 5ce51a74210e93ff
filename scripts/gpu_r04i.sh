#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r04i; mkdir -p $O; cd $R
SB=$R/tools/bin/step_bench
{
for rep in 1 2 3; do
  echo "== launches, in-line grouped wgrad"; timeout 120 $SB --graph 2 --h2d 2 --steps 100 --warmup 20
  echo "== launches, grouped wgrad on the side stream under the next layer's dgrad chain (MB_OVERLAP_WGRAD=1)"; MB_OVERLAP_WGRAD=1 timeout 120 $SB --graph 0 --h2d 0 --steps 100 --warmup 20
  echo "== launches graph 0 in-line"; timeout 120 $SB --graph 0 --h2d 0 --steps 100 --warmup 20
done
} > $O/overlap.txt 2>&1
cut -c1-200 $O/overlap.txt
