#!/bin/bash
# MAG-XLNet attention (staging at once, one flush) + head kernels (block-level reductions): parity and timing
mkdir -p gpurun_out
out=gpurun_out/xl_head.txt
: > $out
timeout 600 python -m pytest tests/test_xlnet_gpu.py tests/test_model_gpu.py -q -x -k "xlnet or gradients_match or mask_replay or fused_training or three_optimizer or edge_shapes" 2>&1 | grep -v "Warning\|^  warn\|^$" | tail -6 >> $out
(timeout 300 python bench.py --model xlnet --cpu-baseline 0 --steps 40 --warmup 8 2>&1 | tail -1 | cut -c1-300) >> $out
for v in 1 2; do timeout 120 tools/bin/step_bench --steps 200 --warmup 30 --graph 1 --h2d 2 >> $out 2>&1; done
cat $out
