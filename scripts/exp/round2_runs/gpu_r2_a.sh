#!/bin/bash
# round 2, visit A: torch-free step timings (eager / prologue+eager / graph), swizzle A/B, overlap A/B, then tests + bench
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
SB=$R/tools/bin/step_bench
{
echo "== eager";            $SB --graph 0 --steps 40 --warmup 8
echo "== prologue+eager";   $SB --graph 2 --steps 40 --warmup 8
echo "== graph";            $SB --graph 1 --steps 40 --warmup 8
echo "== graph h2d";        $SB --graph 1 --h2d 1 --steps 40 --warmup 8
echo "== graph, r1 swizzle"; MB_GEMM_DBG=8 $SB --graph 1 --steps 40 --warmup 8
echo "== graph, new swizzle (again)"; $SB --graph 1 --steps 40 --warmup 8
echo "== graph, r1 swizzle (again)"; MB_GEMM_DBG=8 $SB --graph 1 --steps 40 --warmup 8
echo "== graph, serial wgrad"; MB_OVERLAP_WGRAD=0 $SB --graph 1 --steps 40 --warmup 8
echo "== graph, 4 wgrad launches on side stream"; MB_GROUP_WGRAD=0 $SB --graph 1 --steps 40 --warmup 8
echo "== graph C5 shape (B=32 L=128 V=35)"; $SB --graph 1 --batch 32 --seq 128 --visual 35 --steps 30 --warmup 6
echo "== eager C5 shape"; $SB --graph 0 --batch 32 --seq 128 --visual 35 --steps 30 --warmup 6
} 2>&1 | tee gpurun_out/r2a_step_bench.log
(timeout 1200 python -m pytest tests -m gpu -x -q --tb=short -p no:cacheprovider 2>&1 | tail -40) > gpurun_out/r2a_pytest.log 2>&1
tail -15 gpurun_out/r2a_pytest.log
(timeout 600 python bench.py --steps 30 --warmup 8 --cpu-baseline 0 2>&1 | tail -3) > gpurun_out/r2a_bench.log 2>&1
tail -2 gpurun_out/r2a_bench.log | cut -c1-1500
