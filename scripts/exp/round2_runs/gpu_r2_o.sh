#!/bin/bash
# attention: staged-at-once operands, single bias flush, 8 waves at L = 128 -- parity, phases, step time
mkdir -p gpurun_out
out=gpurun_out/attn_after.txt
: > $out
timeout 400 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -q -x -k "attention or gradients_match or mask_replay or head_mask or c5_shape or training_step_bf16" 2>&1 | tail -8 >> $out
echo "== attention" >> $out
MB_ATTN_TRACE=1 timeout 60 tools/bin/attn_bench >> $out 2>&1
MB_ATTN_TRACE=1 timeout 60 tools/bin/attn_bench --batch 32 --seq 128 >> $out 2>&1
timeout 60 tools/bin/attn_bench >> $out 2>&1
for v in 1 2; do timeout 120 tools/bin/step_bench --steps 200 --warmup 30 --graph 1 --h2d 2 >> $out 2>&1; done
timeout 120 tools/bin/step_bench --steps 100 --warmup 20 --graph 1 --h2d 2 --batch 32 --seq 128 --visual 35 >> $out 2>&1
cat $out
