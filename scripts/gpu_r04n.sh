#!/bin/bash
# round 4, late: XLNet + DP modules on the MAG / pos_emb changes, XLNet bench A/B against the committed HEAD~ library if present
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/r04n; O=gpurun_out/r04n
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_xlnet_gpu.py tests/test_dp_gpu.py -x -q 2>&1 | tail -n 4 > $O/tests.txt
cat $O/tests.txt
for rep in 1 2; do
  for v in "MB_MAG_WGRAD_STAGES=2 MB_MAG_WGRAD_DIRECT=0 MB_PROLOGUE_PACKW=0" "MB_MAG_WGRAD_STAGES=4"; do
    echo "== $v" >> $O/xl.txt
    env $v timeout 200 python bench.py --model xlnet --steps 100 --warmup 15 --cpu-baseline 0 --roofline 0 2>&1 | grep '^{"metric' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])" >> $O/xl.txt
  done
done
cat $O/xl.txt
( cd /tmp && rm -rf /tmp/p_x && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_x -o b -- python $R/bench.py --model xlnet --steps 10 --warmup 3 --cpu-baseline 0 --roofline 0 > /dev/null 2>&1 )
f=$(find /tmp/p_x -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -E "pos_emb|grouped_tn_kernelIDF16bLi64|prologue|unpack|pack_w" $f | cut -c1-60,100-260 | awk -F, '{print $NF, $0}' | cut -c1-200
