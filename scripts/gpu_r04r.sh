#!/bin/bash
# ring geometry of the 128 x 128 grouped weight gradient (MB_GROUP_STAGES: 2 = two 128-byte-row slots (default), 3, 24 | 25 = 4 | 5 slots of 64-byte rows)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/r04r; O=gpurun_out/r04r
ms() { grep -o "[0-9.]* ms/step (events)" | awk '{print $1}'; }
: > $O/group_stages.txt
for rep in 1 2; do for v in 2 24 25; do
  echo "MB_GROUP_STAGES=$v: $(MB_GROUP_STAGES=$v timeout 100 tools/bin/gemm_bench --nset 24 2>&1 | grep 'wgrad x4' | cut -c1-90) | step $(MB_GROUP_STAGES=$v timeout 120 tools/bin/step_bench --steps 200 --warmup 30 --graph 1 --h2d 2 2>&1 | ms) ms" | tee -a $O/group_stages.txt
done; done
