#!/bin/bash
# rocprofv3 PMC passes over the real training step (separate passes, --kernel-trace only): HBM-side traffic per launch of every
# kernel.  Writes gpurun_out/pmc_step/{fetch,write}.csv reduced to per-kernel means by scripts/exp/pmc_reduce.py.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/pmc_step
for set in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcs_$set
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmcs_$set -o p -- python $R/bench.py --steps 6 --warmup 2 --cpu-baseline 0 --roofline 0 > /tmp/pmcs_$set.log 2>&1
  f=$(find /tmp/pmcs_$set -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python $R/scripts/exp/pmc_reduce.py $f $set > $R/gpurun_out/pmc_step/$set.txt; else tail -5 /tmp/pmcs_$set.log; fi
done
cat $R/gpurun_out/pmc_step/*.txt
