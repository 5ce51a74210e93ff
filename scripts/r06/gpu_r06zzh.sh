#!/bin/bash
# final tree: smoke(), the full GPU suite twice (flake check), default bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
export TMPDIR=/tmp
OUT=$R/gpurun_out/r06_gpu_suite_final.txt
{
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
for rep in 1 2; do
  timeout 1200 python -m pytest tests/ -q -m gpu 2>&1 | tail -4
done
timeout 400 python bench.py 2>/dev/null | cut -c1-400
} > $OUT 2>&1
cat $OUT
