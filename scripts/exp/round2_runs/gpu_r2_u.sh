#!/bin/bash
# MAG-XLNet output_attentions + smoke() with the single-call steps + a bench line that reads the refreshed PMC traffic
mkdir -p gpurun_out
out=gpurun_out/xl_attn.txt
: > $out
timeout 400 python -m pytest tests/test_xlnet_gpu.py -q -x -k "attentions or mask_replay or base_model" -s 2>&1 | grep -v "Warning\|^  warn\|^$" | tail -12 >> $out
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -5 >> $out
(timeout 400 python bench.py 2>&1 | tail -1) > gpurun_out/bench_line_final.json
cut -c1-400 gpurun_out/bench_line_final.json >> $out
cat $out
