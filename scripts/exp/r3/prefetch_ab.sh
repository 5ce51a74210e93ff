#!/bin/bash
# piggy-backed weight prefetch (common.h Prefetch): same-box A/B MB_PREFETCH=0 / 1, per-kernel stats of both
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$(pwd); OUT=gpurun_out/r3_prefetch_ab.txt; : > $OUT
S="tools/bin/step_bench --steps 200 --warmup 30 --graph 1 --h2d 2"
for rep in 1 2 3; do
  echo "== MB_PREFETCH=0" >> $OUT; MB_PREFETCH=0 timeout 100 $S | tail -1 >> $OUT
  echo "== MB_PREFETCH=1" >> $OUT; MB_PREFETCH=1 timeout 100 $S | tail -1 >> $OUT
done
for pfv in 0 1; do
  ( cd /tmp && rm -rf /tmp/pp && MB_PREFETCH=$pfv timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -o sb -- $R/tools/bin/step_bench --graph 1 --h2d 2 --steps 25 --warmup 5 > /dev/null 2>&1 )
  f=$(find /tmp/pp -name "*kernel_stats.csv" | head -1)
  echo "== kernel stats MB_PREFETCH=$pfv" >> $OUT
  python3 - $f >> $OUT <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("sum of kernel durations %.1f us" % (tot / 1e3))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:16]:
    print("%6d x %8.2f us  %s" % (int(r["Calls"]), float(r["AverageNs"]) / 1e3, r["Name"][:100]))
PY
done
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -x -q -m gpu -k "layernorm or ln_ or oracle or step_graph" 2>&1 | tail -3 >> $OUT
cat $OUT
