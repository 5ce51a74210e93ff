#!/bin/bash
# One line per gpurun call: which box, its clocks / power cap, and what the HBM-bound AdamW sweep runs at on it (VERDICT r5 item 5b:
# is the 18 % spread of the sweep a property of the box?).  Writes gpurun_out/box_log_last.txt (merged back by gpurun; appended to profiles/r06_box_log.txt by hand after the call).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/box_log_last.txt; rm -f $O
{
  echo "== $(date -u +%H:%M:%S) host $(hostname) gpu $(rocm-smi --showserial 2>/dev/null | grep -i serial | head -1 | awk '{print $NF}')"
  rocm-smi --showclocks --showpower --showmaxpower --showtemp 2>/dev/null | grep -E "sclk|mclk|fclk|Power \(W\)|junction|memory\)" | sed 's/^GPU\[0\]\s*: //' | tr '\n' ';'; echo
  rocm-smi --showcomputepartition --showmemorypartition --showperflevel --showrasinfo 2>/dev/null | grep -E "Partition|partition|Performance|UMC" | sed 's/^GPU\[0\]\s*: //' | tr '\n' ';'; echo
  cat /sys/class/drm/card*/device/current_memory_partition /sys/class/drm/card*/device/current_compute_partition 2>/dev/null | tr '\n' ' '; echo
  timeout 60 $R/tools/bin/adamw_bench 2>&1 | tail -n 3
  # the same sweep again right behind a sustained run (clocks settled)
  timeout 60 $R/tools/bin/step_bench --graph 1 --h2d 2 --steps 300 --warmup 20 2>&1 | grep -o "[0-9.]* ms/step (events)" | head -1
  timeout 60 $R/tools/bin/adamw_bench 2>&1 | tail -n 1
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" | sed 's/^GPU\[0\]\s*: //' | tr '\n' ';'; echo
} >> $O 2>&1
tail -n 9 $O
