#!/bin/bash
# what does the step prologue cost with the batch in pinned host memory vs resident in HBM? (kernel trace of step_bench)
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$(pwd); OUT=gpurun_out/r3_prologue_probe.txt; : > $OUT
for h in 2 0; do
  for pk in 1 0; do
    ( cd /tmp && rm -rf /tmp/pp && MB_PROLOGUE_PACK=$pk timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -o sb -- $R/tools/bin/step_bench --graph 1 --h2d $h --steps 25 --warmup 5 > /dev/null 2>&1 )
    f=$(find /tmp/pp -name "*kernel_stats.csv" | head -1)
    echo "== h2d=$h MB_PROLOGUE_PACK=$pk" >> $OUT
    grep -E "step_prologue|pack_pad|embed_fwd" $f | awk -F, '{print $1, $2, $4}' >> $OUT
    MB_PROLOGUE_PACK=$pk timeout 60 tools/bin/step_bench --graph 1 --h2d $h --steps 200 --warmup 30 | tail -1 >> $OUT
  done
done
cat $OUT
