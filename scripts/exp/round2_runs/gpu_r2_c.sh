#!/bin/bash
# every command under its own short timeout (a hung kernel must not eat the GPU budget)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
SB=$R/tools/bin/step_bench
{
echo "== overlap, launches";  timeout 60 $SB --graph 2 --h2d 2 --steps 40 --warmup 8
echo "== inline, launches";   MB_OVERLAP_WGRAD=0 timeout 60 $SB --graph 2 --h2d 2 --steps 40 --warmup 8
echo "== inline, graph";      MB_OVERLAP_WGRAD=0 timeout 60 $SB --graph 1 --h2d 2 --steps 40 --warmup 8
echo "== overlap, launches";  timeout 60 $SB --graph 2 --h2d 2 --steps 40 --warmup 8
echo "== inline, graph";      MB_OVERLAP_WGRAD=0 timeout 60 $SB --graph 1 --h2d 2 --steps 40 --warmup 8
echo "== inline, launches";   MB_OVERLAP_WGRAD=0 timeout 60 $SB --graph 2 --h2d 2 --steps 40 --warmup 8
} 2>&1 | tee gpurun_out/r2c_step_bench.log
(timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -30) > gpurun_out/r2c_pytest.log 2>&1
tail -8 gpurun_out/r2c_pytest.log
(timeout 120 python scripts/exp/prefetch_diag.py 2>&1 | grep -v Warning | tail -8) > gpurun_out/r2c_diag.log 2>&1
cat gpurun_out/r2c_diag.log
(timeout 300 python bench.py --steps 40 --warmup 8 --cpu-baseline 0 2>&1 | tail -3) > gpurun_out/r2c_bench.log 2>&1
tail -1 gpurun_out/r2c_bench.log | cut -c1-1300
