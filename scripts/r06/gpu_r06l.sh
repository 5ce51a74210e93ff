#!/bin/bash
# round 6: per-kernel in-step durations with the grouped weight gradient on 128 x 128 vs 256 x 128 ping-pong tiles
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/r06l
export TMPDIR=/tmp
for wt in 128 256; do
  ( cd /tmp && rm -rf /tmp/ks_$wt && MB_GROUP_WGRAD=$wt timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$wt -o sb -- $R/tools/bin/step_bench --graph 1 --h2d 2 --steps 25 --warmup 5 2>&1 | grep step_bench ) > gpurun_out/r06l/kstats_$wt.txt
  f=$(find /tmp/ks_$wt -name "*kernel_stats.csv" | head -1)
  cp $f gpurun_out/r06l/kstats_$wt.csv
  g=$(find /tmp/ks_$wt -name "*kernel_trace.csv" | head -1)
  python3 - $f $g >> gpurun_out/r06l/kstats_$wt.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("sum of kernel durations %.1f us over the run" % (tot / 1e3))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:16]:
    print("%6d x %8.2f us  %5.1f%%  %s" % (int(r["Calls"]), float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot, r["Name"][:90]))
# what follows / precedes the grouped wgrad: start-to-start distances
tr = list(csv.DictReader(open(sys.argv[2])))
tr.sort(key=lambda r: int(r["Start_Timestamp"]))
import collections
d = collections.defaultdict(list)
for i in range(1, len(tr) - 1):
    n = tr[i]["Kernel_Name"]
    if "grouped" in n and "256ELi128" in n or "grouped" in n and "128ELi128" in n or "pp_grouped" in n:
        s, e = int(tr[i]["Start_Timestamp"]), int(tr[i]["End_Timestamp"])
        d["dur"].append(e - s)
        d["gap_before"].append(s - int(tr[i - 1]["End_Timestamp"]))
        d["gap_after"].append(int(tr[i + 1]["Start_Timestamp"]) - e)
        d["prev_to_next_start"].append(int(tr[i + 1]["Start_Timestamp"]) - int(tr[i - 1]["End_Timestamp"]))
        d["next_dur:" + tr[i + 1]["Kernel_Name"][:40]].append(int(tr[i + 1]["End_Timestamp"]) - int(tr[i + 1]["Start_Timestamp"]))
for k, v in d.items():
    v.sort()
    print("%-60s n=%4d median %8.2f us  mean %8.2f" % (k, len(v), v[len(v) // 2] / 1e3, sum(v) / len(v) / 1e3))
PY
  echo "== $wt"; cat gpurun_out/r06l/kstats_$wt.txt
done
