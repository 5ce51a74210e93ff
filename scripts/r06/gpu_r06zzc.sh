#!/bin/bash
# ping-pong kernels: DMA pieces split between LOAD and COMP vs all in COMP: parity, loop trace, stand-alone, same-box step A/B
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
OUT=$R/gpurun_out/r06_pp_split_dma.txt
GB=$R/tools/bin/gemm_bench; SB=$R/tools/bin/step_bench
OLD="LD_LIBRARY_PATH=$R/gpurun_ab/nosplit:$LD_LIBRARY_PATH"
bash scripts/box_log.sh > /dev/null 2>&1
{
timeout 300 python -m pytest tests/test_ops_gpu.py -q -x -k "gemm" 2>&1 | tail -3
for only in "fwd ffn2" "dgrad ffn1"; do
    echo "== $only, nset 6 (looptrace build, split DMA)"
    MB_GEMM_TRACE=1 LD_LIBRARY_PATH=$R/gpurun_ab/lt:$LD_LIBRARY_PATH timeout 120 $GB --only "$only" --nset 6 --looptrace 2 2>&1
done
echo "== wgrad 256, nset 6 (looptrace build, split DMA)"
MB_GEMM_TRACE=1 LD_LIBRARY_PATH=$R/gpurun_ab/lt:$LD_LIBRARY_PATH timeout 120 $GB --only wgrad --wtile 256 --nset 6 --looptrace 2 2>&1
for rep in 1 2; do
  echo "== gemm_bench split"; timeout 120 $GB --T 2400 --nset 24 --wtile 256 2>&1 | grep -v "probe"
  echo "== gemm_bench all in COMP"; env $OLD timeout 120 $GB --T 2400 --nset 24 --wtile 256 2>&1 | grep -v "probe"
done
for rep in 1 2 3; do
  echo "== step B=48 L=50 all in COMP"; env $OLD timeout 60 $SB --graph 1 --h2d 2 --steps 100 --warmup 10 2>&1 | grep -o "[0-9.]* ms/step (events)"
  echo "== step B=48 L=50 split"; timeout 60 $SB --graph 1 --h2d 2 --steps 100 --warmup 10 2>&1 | grep -o "[0-9.]* ms/step (events)"
done
for rep in 1 2; do
  echo "== step xlnet all in COMP"; env $OLD timeout 60 $SB --model xlnet --graph 1 --h2d 2 --steps 60 --warmup 10 2>&1 | grep -o "[0-9.]* ms/step (events)"
  echo "== step xlnet split"; timeout 60 $SB --model xlnet --graph 1 --h2d 2 --steps 60 --warmup 10 2>&1 | grep -o "[0-9.]* ms/step (events)"
  echo "== step C5 all in COMP"; env $OLD timeout 60 $SB --graph 1 --h2d 2 --batch 32 --seq 128 --visual 35 --steps 60 --warmup 10 2>&1 | grep -o "[0-9.]* ms/step (events)"
  echo "== step C5 split"; timeout 60 $SB --graph 1 --h2d 2 --batch 32 --seq 128 --visual 35 --steps 60 --warmup 10 2>&1 | grep -o "[0-9.]* ms/step (events)"
done
for cfg in "MB_ADAMW_RIDE_PARAMS=2700000" "MB_ADAMW_RIDE_PARAMS=3000000" "MB_ADAMW_RIDE_ATTN_PARAMS=3000000" "MB_ADAMW_RIDE_ATTN_PARAMS=3600000" "MB_X=0"; do
    echo "== step B=48 L=50 $cfg"; env $cfg timeout 60 $SB --graph 1 --h2d 2 --steps 100 --warmup 10 2>&1 | grep -o "[0-9.]* ms/step (events)"
done
} > $OUT 2>&1
cat $OUT
