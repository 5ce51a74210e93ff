#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r04b; mkdir -p $O; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
SB=$R/tools/bin/step_bench
A="--graph 1 --h2d 2 --steps 100 --warmup 20"
{
for rep in 1 2 3; do
  echo "== single"; timeout 120 $SB $A
  echo "== dp default(4,4,2,2)"; timeout 180 $SB $A --dp 1 2>&1 | grep step_bench
  echo "== dp debug1 (segments only)"; MB_DP_DEBUG=1 timeout 180 $SB $A --dp 1 2>&1 | grep "ms/step"
  echo "== dp debug2 (+events)"; MB_DP_DEBUG=2 timeout 180 $SB $A --dp 1 2>&1 | grep "ms/step"
  echo "== dp debug3 (+allreduce calls, no rows)"; MB_DP_DEBUG=3 timeout 180 $SB $A --dp 1 2>&1 | grep "ms/step"
  echo "== dp chunks 6,4,2"; MB_DP_CHUNKS=6,4,2 timeout 180 $SB $A --dp 1 2>&1 | grep "ms/step"
  echo "== dp chunk 2"; MB_DP_CHUNK=2 timeout 180 $SB $A --dp 1 2>&1 | grep "ms/step"
done
echo "== dp timing on"; timeout 180 $SB $A --dp 1 --timing 1 2>&1 | grep step_bench
echo "== dp launches"; timeout 180 $SB --graph 2 --h2d 2 --steps 100 --warmup 10 --dp 1 2>&1 | grep step_bench
} > $O/dp_step_bench.txt 2>&1
for i in 1 2 3; do timeout 600 python -m pytest tests/test_dp_gpu.py -x -q -k "rccl" > $O/test_rccl_$i.txt 2>&1; done
timeout 1500 python -m pytest tests -m gpu -q > $O/test_all.txt 2>&1
MB_DP_FORCE=1 timeout 300 python bench.py --cpu-baseline 0 --roofline 0 --steps 100 --warmup 20 > $O/bench_dp_force.txt 2>&1
timeout 300 python bench.py --cpu-baseline 0 --roofline 0 --secondary 0 --steps 100 --warmup 20 > $O/bench_plain.txt 2>&1
for f in $O/test_rccl_*.txt $O/test_all.txt; do tail -n 3 $f; done; cat $O/dp_step_bench.txt | cut -c1-200; tail -n 1 $O/bench_dp_force.txt | cut -c1-1800; tail -n 1 $O/bench_plain.txt | cut -c1-900
