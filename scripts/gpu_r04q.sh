#!/bin/bash
# in-step A/B of the ring depth of the 64 x 64 GEMM launches (MB_GEMM_64_STAGES) with and without the k-split kernels (MB_GEMM_KSPLIT)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/r04q; O=gpurun_out/r04q
SB=$R/tools/bin/step_bench
ms() { grep -o "[0-9.]* ms/step (events)" | awk '{print $1}'; }
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -x -q 2>&1 | tail -n 2
echo "== C1: old selection (KS + 2 slots) | 3 slots + KS | 3 slots, no KS (new default) | 4 slots no KS" | tee $O/ab2.txt
for rep in 1 2 3; do
 echo "$(MB_GEMM_64_STAGES=0 MB_GEMM_KSPLIT=1 timeout 120 $SB --steps 200 --warmup 30 --graph 1 --h2d 2 2>&1 | ms) $(MB_GEMM_KSPLIT=1 timeout 120 $SB --steps 200 --warmup 30 --graph 1 --h2d 2 2>&1 | ms) $(timeout 120 $SB --steps 200 --warmup 30 --graph 1 --h2d 2 2>&1 | ms) $(MB_GEMM_64_STAGES=4 timeout 120 $SB --steps 200 --warmup 30 --graph 1 --h2d 2 2>&1 | ms)" | tee -a $O/ab2.txt
done
echo "== C5 (B=32 L=128 V=35): old selection | new default | 4 slots" | tee -a $O/ab2.txt
C5="--steps 100 --warmup 20 --graph 1 --h2d 2 --batch 32 --seq 128 --visual 35"
for rep in 1 2 3; do
 echo "$(MB_GEMM_64_STAGES=0 MB_GEMM_KSPLIT=1 timeout 120 $SB $C5 2>&1 | ms) $(timeout 120 $SB $C5 2>&1 | ms) $(MB_GEMM_64_STAGES=4 timeout 120 $SB $C5 2>&1 | ms)" | tee -a $O/ab2.txt
done
echo "== MAG-XLNet bench.py: old selection | new default" | tee -a $O/ab2.txt
x() { env "$@" timeout 200 python bench.py --model xlnet --steps 100 --warmup 15 --cpu-baseline 0 --roofline 0 2>&1 | grep '^{"metric' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for rep in 1 2; do echo "$(x MB_GEMM_64_STAGES=0 MB_GEMM_KSPLIT=1) $(x A=1)" | tee -a $O/ab2.txt; done
