#!/bin/bash
# kernel trace + stats of the C++ step (tag = $1, extra step_bench args = rest)
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-x}; shift
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o sb -- $R/tools/bin/step_bench --graph 2 --h2d 2 --steps 20 --warmup 5 "$@" 2>&1 | grep step_bench
for f in $(find /tmp/prof -name "*kernel_stats.csv"); do cp $f $R/gpurun_out/${TAG}_kernel_stats.csv; done
for f in $(find /tmp/prof -name "*kernel_trace.csv"); do python3 $R/scripts/exp/trace_gaps.py $f 6 | grep -v "gap .* between" | cut -c1-150 > $R/gpurun_out/${TAG}_trace_gaps.txt; done
head -34 $R/gpurun_out/${TAG}_trace_gaps.txt
