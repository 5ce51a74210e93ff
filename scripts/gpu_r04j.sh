#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r04j; mkdir -p $O; cd $R
export TMPDIR=/tmp
J='^{"metric'
timeout 900 python -m pytest tests/test_xlnet_gpu.py -x -q > $O/test_xlnet.txt 2>&1
{
for rep in 1 2; do
  echo "== base (round-3 XLNet attention kernels)"; LD_LIBRARY_PATH=$R/gpurun_ab/xlbase:$LD_LIBRARY_PATH MB_LIB_DIR=$R/gpurun_ab/xlbase timeout 300 python bench.py --model xlnet --cpu-baseline 0 --roofline 0 --steps 60 --warmup 10 2>&1 | grep "$J" | cut -c1-330
  echo "== new"; timeout 300 python bench.py --model xlnet --cpu-baseline 0 --roofline 0 --steps 60 --warmup 10 2>&1 | grep "$J" | cut -c1-330
done
echo "== new, L=128 B=32"; timeout 300 python bench.py --model xlnet --seq 128 --batch 32 --cpu-baseline 0 --roofline 0 --steps 40 --warmup 8 2>&1 | grep "$J" | cut -c1-330
echo "== base, L=128 B=32"; MB_LIB_DIR=$R/gpurun_ab/xlbase timeout 300 python bench.py --model xlnet --seq 128 --batch 32 --cpu-baseline 0 --roofline 0 --steps 40 --warmup 8 2>&1 | grep "$J" | cut -c1-330
} > $O/xlnet_ab.txt 2>&1
( cd /tmp && rm -rf /tmp/p_x && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_x -o b -- python $R/bench.py --model xlnet --steps 10 --warmup 3 --cpu-baseline 0 --roofline 0 > /dev/null 2>&1 )
f=$(find /tmp/p_x -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/xlnet_kernel_stats.csv
tail -n 5 $O/test_xlnet.txt; cat $O/xlnet_ab.txt; grep -E "xl_attn" $O/xlnet_kernel_stats.csv | cut -c1-60,140-240
