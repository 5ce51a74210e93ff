#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out; out=gpurun_out/r3_xl_check.txt; : > $out
timeout 600 python -m pytest tests/test_xlnet_gpu.py -x -q -m gpu 2>&1 | tail -5 >> $out
for v in 1 0 1 0; do echo "== MB_XL_FUSE_QKV=$v" >> $out; MB_XL_FUSE_QKV=$v timeout 200 python bench.py --model xlnet --cpu-baseline 0 --roofline 0 --steps 40 --warmup 8 2>&1 | tail -1 | cut -c1-200 >> $out; done
cat $out
