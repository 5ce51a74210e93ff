#!/bin/bash
# same-box A/B of the current library against the one of commit a1449b6 (the round's mid-point: ping-pong tile + weight-gradient riders), both shapes
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
bash scripts/box_log.sh > /dev/null 2>&1
SB=$R/tools/bin/step_bench
for rep in 1 2 3; do
  echo "== now,     B=48 L=50";  timeout 60 $SB --graph 1 --h2d 2 --steps 300 --warmup 20 2>&1 | grep -o "[0-9.]* ms/step (events)"
  echo "== a1449b6, B=48 L=50";  LD_LIBRARY_PATH=$R/gpurun_ab/a1449b6:$LD_LIBRARY_PATH timeout 60 $SB --graph 1 --h2d 2 --steps 300 --warmup 20 2>&1 | grep -o "[0-9.]* ms/step (events)"
  echo "== now,     B=32 L=128"; timeout 60 $SB --graph 1 --h2d 2 --batch 32 --seq 128 --visual 35 --steps 150 --warmup 10 2>&1 | grep -o "[0-9.]* ms/step (events)"
  echo "== a1449b6, B=32 L=128"; LD_LIBRARY_PATH=$R/gpurun_ab/a1449b6:$LD_LIBRARY_PATH timeout 60 $SB --graph 1 --h2d 2 --batch 32 --seq 128 --visual 35 --steps 150 --warmup 10 2>&1 | grep -o "[0-9.]* ms/step (events)"
  echo "== now,     xlnet";      timeout 60 $SB --model xlnet --graph 1 --h2d 2 --steps 150 --warmup 10 2>&1 | grep -o "[0-9.]* ms/step (events)"
  echo "== a1449b6, xlnet";      LD_LIBRARY_PATH=$R/gpurun_ab/a1449b6:$LD_LIBRARY_PATH timeout 60 $SB --model xlnet --graph 1 --h2d 2 --steps 150 --warmup 10 2>&1 | grep -o "[0-9.]* ms/step (events)"
done
ldd $SB | grep magbert
