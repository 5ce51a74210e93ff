#!/bin/bash
# per-kernel in-step averages (rocprofv3 --kernel-trace --stats of the torch-free step driver) for the in-tree build and for
# the builds under gpurun_ab/<name> given as arguments; one table per build in gpurun_out/kstats_<name>.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
export TMPDIR=/tmp
ARGS=${ARGS:---graph 1 --h2d 2 --steps 25 --warmup 5}
for v in current "$@"; do
  LP=$LD_LIBRARY_PATH; [ "$v" != current ] && LP=$R/gpurun_ab/$v:$LD_LIBRARY_PATH
  ( cd /tmp && rm -rf /tmp/ks_$v && LD_LIBRARY_PATH=$LP timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$v -o sb -- $R/tools/bin/step_bench $ARGS 2>&1 | grep step_bench ) > gpurun_out/kstats_$v.txt
  f=$(find /tmp/ks_$v -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f gpurun_out/kstats_$v.csv && python3 - $f >> gpurun_out/kstats_$v.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("sum of kernel durations %.1f us over the run" % (tot / 1e3))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:34]:
    print("%6d x %8.2f us  %5.1f%%  %s" % (int(r["Calls"]), float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot, r["Name"][:96]))
PY
  echo "== $v"; cat gpurun_out/kstats_$v.txt
done
