#!/bin/bash
# experiment: touch the grouped weight gradient's forward-activation operands (g, y1, x_l: 22 MB from HBM) in front of the launch -- does the
# launch get faster by more than the touches cost?  (step timeline by call site, riders off so that the launch is the tiles alone)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
bash scripts/box_log.sh > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
for pf in 0 128 64; do
  rm -rf /tmp/prof
  MB_PF_WGRAD=$pf MB_ADAMW_RIDE=0 MB_GEMM_LOG=1 timeout 120 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -o sb -- $R/tools/bin/step_bench --graph 1 --h2d 2 --steps 25 --warmup 5 2> /tmp/gl.txt | grep -o "[0-9.]* ms/step (events)"
  f=$(find /tmp/prof -name "*kernel_trace.csv" | head -1)
  python3 $R/scripts/exp/step_timeline.py $f /tmp/gl.txt 10 > $R/gpurun_out/r06_timeline_pf$pf.txt
  echo "== MB_PF_WGRAD=$pf: grouped weight gradient launches, touch launches (avg us)"
  grep "gemm_pp_grouped" $R/gpurun_out/r06_timeline_pf$pf.txt | awk '{s+=$3; n++} END {print "wgrad", n, s/n}'
  grep "touch_kernel" $R/gpurun_out/r06_timeline_pf$pf.txt | awk '{s+=$3; n++} END {if (n) print "touch", n, s/n}'
  grep "attn_bwd" $R/gpurun_out/r06_timeline_pf$pf.txt | awk '{s+=$3; n++} END {print "attn_bwd", n, s/n}'
done
for rep in 1 2; do for pf in 0 128; do echo "== untraced MB_PF_WGRAD=$pf"; MB_PF_WGRAD=$pf timeout 60 $R/tools/bin/step_bench --graph 1 --h2d 2 --steps 300 --warmup 20 2>&1 | grep -o "[0-9.]* ms/step (events)"; done; done
