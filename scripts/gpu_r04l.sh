#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r04l; mkdir -p $O; cd $R
J='^{"metric'
timeout 900 python -m pytest tests/test_xlnet_gpu.py -q > $O/test_xlnet.txt 2>&1
{
for rep in 1 2; do
  echo "== MB_XL_SPLIT_R=0"; MB_XL_SPLIT_R=0 timeout 300 python bench.py --model xlnet --cpu-baseline 0 --roofline 0 --steps 60 --warmup 10 2>&1 | grep "$J" | cut -c1-330
  echo "== MB_XL_SPLIT_R=1 (default)"; timeout 300 python bench.py --model xlnet --cpu-baseline 0 --roofline 0 --steps 60 --warmup 10 2>&1 | grep "$J" | cut -c1-330
done
} > $O/xlnet_split_r_ab.txt 2>&1
tail -n 5 $O/test_xlnet.txt; cat $O/xlnet_split_r_ab.txt
