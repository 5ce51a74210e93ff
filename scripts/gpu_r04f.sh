#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r04f; mkdir -p $O; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
J='^{"metric'
{
echo "== DP fp32 wire"; MB_DP_FORCE=1 MB_DP_GRAD_DTYPE=fp32 timeout 300 python bench.py --cpu-baseline 0 --roofline 0 --steps 100 --warmup 20 2>&1 | grep "$J"
echo "== DP bf16 wire, normal priority comm stream"; MB_DP_FORCE=1 MB_DP_COMM_PRIORITY=0 timeout 300 python bench.py --cpu-baseline 0 --roofline 0 --steps 100 --warmup 20 2>&1 | grep "$J"
echo "== DP fp32 wire, normal priority"; MB_DP_FORCE=1 MB_DP_GRAD_DTYPE=fp32 MB_DP_COMM_PRIORITY=0 timeout 300 python bench.py --cpu-baseline 0 --roofline 0 --steps 100 --warmup 20 2>&1 | grep "$J"
echo "== DP fp32 wire, dense emb"; MB_DP_FORCE=1 MB_DP_GRAD_DTYPE=fp32 MB_DP_SPARSE_EMB=0 timeout 300 python bench.py --cpu-baseline 0 --roofline 0 --steps 100 --warmup 20 2>&1 | grep "$J"
} > $O/dp_variants.txt 2>&1
( cd /tmp && rm -rf /tmp/p_dp && MB_DP_FORCE=1 MB_DP_GRAD_DTYPE=fp32 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_dp -o dp -- python $R/bench.py --cpu-baseline 0 --roofline 0 --steps 20 --warmup 5 > $O/prof_stdout.txt 2>&1 )
f=$(find /tmp/p_dp -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -40 $f > $O/dp_kernel_stats.csv
cut -c1-400 $O/dp_variants.txt; cut -c1-200 $O/dp_kernel_stats.csv | head -30
