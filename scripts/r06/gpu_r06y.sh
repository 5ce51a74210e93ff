#!/bin/bash
# in-step timeline by call site (profiles/r06_step_timeline.txt)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof
MB_GEMM_LOG=1 timeout 120 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -o sb -- $R/tools/bin/step_bench --graph 1 --h2d 2 --steps 25 --warmup 5 2> /tmp/gl.txt | grep step_bench
grep -c "magbert gemm" /tmp/gl.txt
for f in $(find /tmp/prof -name "*kernel_trace.csv"); do python3 $R/scripts/exp/step_timeline.py $f /tmp/gl.txt 10 > $R/gpurun_out/r06_step_timeline.txt; done
tail -3 $R/gpurun_out/r06_step_timeline.txt
