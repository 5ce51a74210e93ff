#!/bin/bash
# the touches of the weight gradient's forward-activation operands carried by the attention backward (MB_PF_WGRAD): timeline + untraced A/B
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
bash scripts/box_log.sh > /dev/null 2>&1
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -x -q -k "attention or gradients or riding" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
for pf in 0 128; do
  rm -rf /tmp/prof
  MB_PF_WGRAD=$pf MB_GEMM_LOG=1 timeout 120 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -o sb -- $R/tools/bin/step_bench --graph 1 --h2d 2 --steps 25 --warmup 5 2> /tmp/gl.txt | grep -o "[0-9.]* ms/step (events)"
  f=$(find /tmp/prof -name "*kernel_trace.csv" | head -1)
  python3 $R/scripts/exp/step_timeline.py $f /tmp/gl.txt 10 > $R/gpurun_out/r06_timeline_pf$pf.txt
  echo "== traced, MB_PF_WGRAD=$pf (avg us per launch)"
  grep "gemm_pp_grouped" $R/gpurun_out/r06_timeline_pf$pf.txt | awk '{s+=$3; n++} END {print "wgrad", n, s/n}'
  grep "attn_bwd" $R/gpurun_out/r06_timeline_pf$pf.txt | awk '{s+=$3; n++} END {print "attn_bwd", n, s/n}'; grep "gemm2_ride" $R/gpurun_out/r06_timeline_pf$pf.txt | awk '{s+=$3; n++} END {if (n) print "dgrad with riders", n, s/n}'; grep "gemm2_kernel<128,128,0,1,4" $R/gpurun_out/r06_timeline_pf$pf.txt | awk '{s+=$3; n++} END {if (n) print "dgrad ffn2", n, s/n}'
done
for rep in 1 2 3; do for pf in 0 128 64 256; do echo "== untraced MB_PF_WGRAD=$pf"; MB_PF_WGRAD=$pf timeout 60 $R/tools/bin/step_bench --graph 1 --h2d 2 --steps 300 --warmup 20 2>&1 | grep -o "[0-9.]* ms/step (events)"; done; done
for rep in 1 2; do for pf in 0 128; do echo "== C5 untraced MB_PF_WGRAD=$pf"; MB_PF_WGRAD=$pf timeout 60 $R/tools/bin/step_bench --graph 1 --h2d 2 --batch 32 --seq 128 --visual 35 --steps 100 --warmup 10 2>&1 | grep -o "[0-9.]* ms/step (events)"; done; done
