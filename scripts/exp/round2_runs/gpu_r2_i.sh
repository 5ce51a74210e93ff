#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
{
for v in 1 0 1 0; do echo "== GROUP_MAP=$v"; MB_GROUP_MAP=$v timeout 60 tools/bin/gemm_bench --only wgrad; MB_GROUP_MAP=$v timeout 60 tools/bin/step_bench --graph 1 --h2d 2 --steps 40 --warmup 8; done
} 2>&1 | tee gpurun_out/r2i.log
export TMPDIR=/tmp
for v in 1 0; do
( cd /tmp && rm -rf /tmp/p_f$v && MB_GROUP_MAP=$v timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/p_f$v -o p -- $R/tools/bin/step_bench --graph 2 --h2d 2 --steps 6 --warmup 2 > /dev/null 2>&1 )
f=$(find /tmp/p_f$v -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python3 scripts/exp/pmc_reduce.py $f FETCH_SIZE | grep grouped
done
(timeout 500 python -m pytest tests/test_ops_gpu.py tests/test_xlnet_gpu.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -4)
