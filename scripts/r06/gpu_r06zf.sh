#!/bin/bash
# riders in the ffn2 dgrad launch too (MB_ADAMW_RIDE_DGRAD=2: 128 x 128 tiles, 56 half-idle CUs)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
bash scripts/box_log.sh > /dev/null 2>&1
MB_ADAMW_RIDE_DGRAD=2 timeout 600 python -m pytest tests/test_model_gpu.py -x -q -k "riding" 2>&1 | tail -2
for rep in 1 2 3; do
  for cfg in "MB_ADAMW_RIDE_DGRAD=0" "MB_ADAMW_RIDE_DGRAD=1" "MB_ADAMW_RIDE_DGRAD=2" "MB_ADAMW_RIDE_DGRAD=2 MB_ADAMW_RIDE_DGELU_PARAMS=400000" "MB_ADAMW_RIDE_DGRAD=2 MB_ADAMW_RIDE_DGELU_PARAMS=1200000"; do
    echo "== $cfg"; env $cfg timeout 60 $R/tools/bin/step_bench --graph 1 --h2d 2 --steps 300 --warmup 20 2>&1 | grep -o "[0-9.]* ms/step (events)"
  done
done
cd /tmp && export TMPDIR=/tmp
for cfg in "MB_ADAMW_RIDE_DGRAD=1" "MB_ADAMW_RIDE_DGRAD=2"; do
  rm -rf /tmp/prof
  env $cfg MB_GEMM_LOG=1 timeout 120 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -o sb -- $R/tools/bin/step_bench --graph 1 --h2d 2 --steps 25 --warmup 5 2> /tmp/gl.txt | grep -o "[0-9.]* ms/step (events)"
  f=$(find /tmp/prof -name "*kernel_trace.csv" | head -1)
  python3 $R/scripts/exp/step_timeline.py $f /tmp/gl.txt 10 > /tmp/tl.txt
  echo "== traced, $cfg (avg us per launch)"
  for k in "gemm_pp_grouped" "gemm2_ride_kernel<64" "gemm2_ride_kernel<128" "gemm2_kernel<128,128,0,1,4" "adamw"; do grep "$k" /tmp/tl.txt | awk -v k="$k" '{s+=$3; n++} END {if (n) print k, n, s/n}'; done
done
