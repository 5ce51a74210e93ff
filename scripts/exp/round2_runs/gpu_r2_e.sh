#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
SB=$R/tools/bin/step_bench
{
for v in 1 0 1 0; do echo "== KSPLIT=$v"; MB_GEMM_KSPLIT=$v timeout 60 $SB --graph 1 --h2d 2 --steps 40 --warmup 8; done
echo "== C5 KSPLIT=1"; MB_GEMM_KSPLIT=1 timeout 60 $SB --graph 1 --h2d 2 --batch 32 --seq 128 --visual 35 --steps 30 --warmup 6
echo "== C5 KSPLIT=0"; MB_GEMM_KSPLIT=0 timeout 60 $SB --graph 1 --h2d 2 --batch 32 --seq 128 --visual 35 --steps 30 --warmup 6
} 2>&1 | tee gpurun_out/r2e_step_bench.log
(timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider 2>&1 | tail -15) > gpurun_out/r2e_pytest.log 2>&1
tail -6 gpurun_out/r2e_pytest.log
