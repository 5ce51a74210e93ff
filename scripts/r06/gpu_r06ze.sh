#!/bin/bash
# what do rider workgroups cost a 64 x 64 dgrad launch when they do (almost) nothing?  MB_PF_WGRAD=<huge stride> = one touch per region
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
cd /tmp && export TMPDIR=/tmp
for cfg in "MB_PF_WGRAD=0 MB_ADAMW_RIDE_DGRAD=0" "MB_PF_WGRAD=100000000 MB_ADAMW_RIDE_DGRAD=0" "MB_PF_WGRAD=100000000 MB_ADAMW_RIDE_DGRAD=0 MB_ADAMW_RIDE_DGRAD_BLOCKS=8" "MB_PF_WGRAD=128 MB_ADAMW_RIDE_DGRAD=0 MB_ADAMW_RIDE_DGRAD_BLOCKS=8" "MB_PF_WGRAD=1024 MB_ADAMW_RIDE_DGRAD=0"; do
  rm -rf /tmp/prof
  env $cfg MB_GEMM_LOG=1 timeout 120 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -o sb -- $R/tools/bin/step_bench --graph 1 --h2d 2 --steps 25 --warmup 5 2> /tmp/gl.txt | grep -o "[0-9.]* ms/step (events)"
  f=$(find /tmp/prof -name "*kernel_trace.csv" | head -1)
  python3 $R/scripts/exp/step_timeline.py $f /tmp/gl.txt 10 > /tmp/tl.txt
  echo "== traced, $cfg (avg us per launch)"
  for k in "gemm_pp_grouped" "gemm2_ride" "gemm2_kernel<128,128,0,1,4" "gemm2_kernel<64,64,0,1,3"; do grep "$k" /tmp/tl.txt | awk -v k="$k" '{s+=$3; n++} END {if (n) print k, n, s/n}'; done
  grep "gemm2_ride" /tmp/tl.txt | head -4
done
