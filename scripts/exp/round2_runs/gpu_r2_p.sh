#!/bin/bash
# DPP reductions, up-front loads in ln_bwd / embed_pos_type, attention bias flush: parity + timing
mkdir -p gpurun_out
out=gpurun_out/rowops_after.txt
: > $out
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -q -x -k "not driver and not checkpoint and not from_pretrained" 2>&1 | tail -6 >> $out
echo "== attention" >> $out
MB_ATTN_TRACE=1 timeout 60 tools/bin/attn_bench >> $out 2>&1
for v in 1 2; do timeout 120 tools/bin/step_bench --steps 200 --warmup 30 --graph 1 --h2d 2 >> $out 2>&1; done
timeout 120 tools/bin/step_bench --steps 100 --warmup 20 --graph 1 --h2d 2 --batch 32 --seq 128 --visual 35 >> $out 2>&1
cat $out
