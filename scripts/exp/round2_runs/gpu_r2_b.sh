#!/bin/bash
# round 2, visit B: zero-copy batch gather, full GPU tests, bench, kernel trace of the C++ step
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
SB=$R/tools/bin/step_bench
{
echo "== single call, resident";  $SB --graph 2 --steps 40 --warmup 8
echo "== single call, zero-copy"; $SB --graph 2 --h2d 2 --steps 40 --warmup 8
echo "== single call, memcpy";    $SB --graph 2 --h2d 1 --steps 40 --warmup 8
echo "== single call, resident";  $SB --graph 2 --steps 40 --warmup 8
echo "== single call, zero-copy"; $SB --graph 2 --h2d 2 --steps 40 --warmup 8
} 2>&1 | tee gpurun_out/r2b_step_bench.log
(timeout 1200 python -m pytest tests -m gpu -x -q --tb=short -p no:cacheprovider 2>&1 | tail -30) > gpurun_out/r2b_pytest.log 2>&1
tail -8 gpurun_out/r2b_pytest.log
(timeout 300 python scripts/exp/prefetch_diag.py 2>&1 | grep -v Warning | tail -14) > gpurun_out/r2b_diag.log 2>&1
cat gpurun_out/r2b_diag.log
(timeout 600 python bench.py --steps 40 --warmup 8 --cpu-baseline 0 2>&1 | tail -3) > gpurun_out/r2b_bench.log 2>&1
tail -1 gpurun_out/r2b_bench.log | cut -c1-1200
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o sb -- $SB --graph 2 --steps 20 --warmup 5 > /dev/null 2>&1
for f in $(find /tmp/prof -name "*kernel_stats.csv"); do cp $f $R/gpurun_out/r2b_kernel_stats.csv; done
for f in $(find /tmp/prof -name "*kernel_trace.csv"); do python $R/scripts/exp/trace_gaps.py $f 6 | grep -v "gap .* between" | cut -c1-150 > $R/gpurun_out/r2b_trace_gaps.txt; done
head -45 $R/gpurun_out/r2b_trace_gaps.txt
