#!/bin/bash
# round 5, VERDICT r4 item 6: AdamW inside the grouped weight-gradient epilogue (MB_ADAMW_IN_WGRAD=1) against the default step, same box,
# tools/step_bench (one GPU, bf16, B=48 L=50), three times each + the kernel trace of both.  -> gpurun_out/r05/adamw_in_wgrad_ab.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05; mkdir -p $O; cd $R
SB=$R/tools/bin/step_bench
export TMPDIR=/tmp
{
for rep in 1 2 3; do
  echo "== default (gradient stored, AdamW sweeps all 110.9 M parameters)"; timeout 120 $SB --graph 1 --h2d 2 --steps 300 --warmup 30 | grep ms/step
  echo "== MB_ADAMW_IN_WGRAD=1 (85 M parameters updated by the weight-gradient launches)"; MB_ADAMW_IN_WGRAD=1 timeout 120 $SB --graph 1 --h2d 2 --steps 300 --warmup 30 | grep ms/step
done
for v in 0 1; do
  echo "== kernel trace, MB_ADAMW_IN_WGRAD=$v (us per launch, launches per step, ms per step; last 8 steps)"
  ( cd /tmp && rm -rf /tmp/p_aw$v && MB_ADAMW_IN_WGRAD=$v timeout 120 rocprofv3 --kernel-trace --output-format csv -d /tmp/p_aw$v -o sb -- $SB --graph 1 --h2d 2 --steps 12 --warmup 4 > /dev/null 2>&1 )
  f=$(find /tmp/p_aw$v -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python3 scripts/exp/instep_json.py $f 8 "bert B=48 L=50 bf16" /tmp/aw$v.json | grep -i "grouped_tn\|adamw\|busy_ms"
done
} > $O/adamw_in_wgrad_ab.txt 2>&1
cat $O/adamw_in_wgrad_ab.txt
