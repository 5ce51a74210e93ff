#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r04e; mkdir -p $O; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
J='^{"metric'
{
for rep in 1 2; do
  echo "== plain (mb_bert_train_step)"; timeout 300 python bench.py --cpu-baseline 0 --roofline 0 --secondary 0 --steps 100 --warmup 20 2>&1 | grep "$J"
  echo "== MB_DP_FORCE=1 (one-rank RCCL group, mb_bert_train_step_dp)"; MB_DP_FORCE=1 timeout 300 python bench.py --cpu-baseline 0 --roofline 0 --steps 100 --warmup 20 2>&1 | grep "$J"
done
echo "== MB_DP_FORCE=1 MB_DP_ENGINE=0 (round-3 structure: passes and exchange driven from Python)"; MB_DP_FORCE=1 MB_DP_ENGINE=0 timeout 300 python bench.py --cpu-baseline 0 --roofline 0 --steps 100 --warmup 20 2>&1 | grep "$J"
echo "== xlnet plain"; timeout 300 python bench.py --model xlnet --cpu-baseline 0 --roofline 0 --steps 60 --warmup 10 2>&1 | grep "$J"
echo "== xlnet MB_DP_FORCE=1"; MB_DP_FORCE=1 timeout 300 python bench.py --model xlnet --cpu-baseline 0 --roofline 0 --steps 60 --warmup 10 2>&1 | grep "$J"
} > $O/dp_force.txt 2>&1
cut -c1-1300 $O/dp_force.txt
