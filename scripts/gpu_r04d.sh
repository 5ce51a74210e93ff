#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r04d; mkdir -p $O; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_dp_gpu.py -x -q > $O/test_dp.txt 2>&1
{
for rep in 1 2; do
  echo "== plain"; timeout 300 python bench.py --cpu-baseline 0 --roofline 0 --secondary 0 --steps 100 --warmup 20 2>&1 | tail -n 1
  echo "== MB_DP_FORCE=1 (one-rank RCCL group, mb_bert_train_step_dp)"; MB_DP_FORCE=1 timeout 300 python bench.py --cpu-baseline 0 --roofline 0 --steps 100 --warmup 20 2>&1 | tail -n 1
done
echo "== MB_DP_FORCE=1 MB_DP_ENGINE=0 (round-3 structure: passes and exchange driven from Python)"; MB_DP_FORCE=1 MB_DP_ENGINE=0 timeout 300 python bench.py --cpu-baseline 0 --roofline 0 --steps 100 --warmup 20 2>&1 | tail -n 1
echo "== xlnet plain"; timeout 300 python bench.py --model xlnet --cpu-baseline 0 --roofline 0 --steps 60 --warmup 10 2>&1 | tail -n 1
echo "== xlnet MB_DP_FORCE=1"; MB_DP_FORCE=1 timeout 300 python bench.py --model xlnet --cpu-baseline 0 --roofline 0 --steps 60 --warmup 10 2>&1 | tail -n 1
} > $O/dp_force.txt 2>&1
tail -n 3 $O/test_dp.txt; cut -c1-1500 $O/dp_force.txt
