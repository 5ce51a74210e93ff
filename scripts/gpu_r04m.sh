#!/bin/bash
# round 4, late: MAG family (deeper ring for the grouped weight gradient, direct stores, weight pack in the prologue) + q|k|v touch
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/r04m; O=gpurun_out/r04m
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -x -q 2>&1 | tail -n 5 > $O/tests.txt
cat $O/tests.txt
SB=$R/tools/bin/step_bench
run() { # name, env...
  name=$1; shift
  for rep in 1 2; do echo "== $name" >> $O/ab.txt; env "$@" timeout 120 $SB --steps 200 --warmup 30 --graph 1 --h2d 2 >> $O/ab.txt 2>&1; done
}
: > $O/ab.txt
for rep in 1 2; do
run base MB_MAG_WGRAD_STAGES=2 MB_MAG_WGRAD_DIRECT=0 MB_PROLOGUE_PACKW=0 MB_PF_QKV=0
run stages4 MB_MAG_WGRAD_STAGES=4 MB_MAG_WGRAD_DIRECT=0 MB_PROLOGUE_PACKW=0 MB_PF_QKV=0
run stages4_direct MB_MAG_WGRAD_STAGES=4 MB_MAG_WGRAD_DIRECT=1 MB_PROLOGUE_PACKW=0 MB_PF_QKV=0
run all MB_MAG_WGRAD_STAGES=4 MB_MAG_WGRAD_DIRECT=1 MB_PROLOGUE_PACKW=1 MB_PF_QKV=0
run all_qkv64 MB_MAG_WGRAD_STAGES=4 MB_MAG_WGRAD_DIRECT=1 MB_PROLOGUE_PACKW=1 MB_PF_QKV=64
run all_qkv128 MB_MAG_WGRAD_STAGES=4 MB_MAG_WGRAD_DIRECT=1 MB_PROLOGUE_PACKW=1 MB_PF_QKV=128
done
grep -B1 "ms/step" $O/ab.txt | grep -v "^--" | paste - - | awk '{print $2, $12}' | sort | awk '{a[$1]=a[$1]" "$2} END{for(k in a) print k, a[k]}' | sort > $O/ab_summary.txt
cat $O/ab_summary.txt
# kernel tables for the stage variants
kt() { name=$1; shift
  ( cd /tmp && rm -rf /tmp/ks_$name && env "$@" timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$name -o sb -- $SB --graph 1 --h2d 2 --steps 25 --warmup 5 > /dev/null 2>&1 )
  f=$(find /tmp/ks_$name -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && python3 - $f > $O/kstats_$name.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
    n = r["Name"]
    if any(k in n for k in ("grouped_tn_kernelIDF16bLi64", "mag_", "prologue", "attn_bwd", "ln_bwd", "embed", "ln_reduce")):
        print("%6d x %8.2f us  %s" % (int(r["Calls"]), float(r["AverageNs"]) / 1e3, n[:90]))
PY
  echo "== $name"; cat $O/kstats_$name.txt
}
kt base MB_MAG_WGRAD_STAGES=2 MB_MAG_WGRAD_DIRECT=0 MB_PROLOGUE_PACKW=0 MB_PF_QKV=0
kt st3 MB_MAG_WGRAD_STAGES=3 MB_MAG_WGRAD_DIRECT=1 MB_PROLOGUE_PACKW=1 MB_PF_QKV=0
kt st4 MB_MAG_WGRAD_STAGES=4 MB_MAG_WGRAD_DIRECT=1 MB_PROLOGUE_PACKW=1 MB_PF_QKV=0
kt st5 MB_MAG_WGRAD_STAGES=5 MB_MAG_WGRAD_DIRECT=1 MB_PROLOGUE_PACKW=1 MB_PF_QKV=64
