#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench, rocprofv3 kernel stats.  Usage (from the repo root on the box):
#   bash scripts/gpu_round.sh [tests|notests] [tag]
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${2:-r01}
mkdir -p $R/gpurun_out
cd $R
if [ "${1:-tests}" = "tests" ]; then
  (timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -120) > gpurun_out/pytest_gpu.log 2>&1
  tail -5 gpurun_out/pytest_gpu.log
  (timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -20) > gpurun_out/smoke.log 2>&1
  tail -3 gpurun_out/smoke.log
fi
(timeout 900 python bench.py 2>&1 | tail -5) > gpurun_out/bench_${TAG}.log 2>&1
tail -2 gpurun_out/bench_${TAG}.log
(timeout 600 python bench.py --model xlnet --cpu-steps 1 2>&1 | tail -2) > gpurun_out/bench_xlnet_${TAG}.log 2>&1
tail -1 gpurun_out/bench_xlnet_${TAG}.log | cut -c1-400
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof && mkdir -p /tmp/prof
(timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python $R/bench.py --steps 10 --warmup 3 --cpu-baseline 0 2>&1 | tail -3) > $R/gpurun_out/rocprof_${TAG}.log 2>&1
find /tmp/prof -type f | head
for f in $(find /tmp/prof -name "*kernel_stats.csv"); do cp $f $R/gpurun_out/bench_kernel_stats_${TAG}.csv; done
head -40 $R/gpurun_out/bench_kernel_stats_${TAG}.csv
