#!/bin/bash
# round 4, first GPU visit: the new tests, the whole GPU suite, the single-call data-parallel step against the single-call step on
# the same box (C++ driver and bench.py), the secondary workloads.  Every command under its own timeout.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r04a; mkdir -p $O; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
SB=$R/tools/bin/step_bench
{
for rep in 1 2; do
  timeout 120 $SB --graph 1 --h2d 2 --steps 200 --warmup 30
  timeout 180 $SB --graph 1 --h2d 2 --steps 200 --warmup 30 --dp 1
  MB_DP_CHUNK=1 timeout 180 $SB --graph 1 --h2d 2 --steps 200 --warmup 30 --dp 1
  MB_DP_CHUNK=3 timeout 180 $SB --graph 1 --h2d 2 --steps 200 --warmup 30 --dp 1
done
timeout 180 $SB --graph 1 --h2d 2 --steps 200 --warmup 30 --dp 1 --sparse 0
timeout 180 $SB --graph 1 --h2d 2 --steps 200 --warmup 30 --dp 1 --wire bf16
timeout 180 $SB --graph 2 --h2d 2 --steps 100 --warmup 10 --dp 1
} > $O/dp_step_bench.txt 2>&1
timeout 900 python -m pytest tests/test_dp_gpu.py -x -q -k "single_call or row_exchange" > $O/test_dp_new.txt 2>&1
timeout 600 python -m pytest tests/test_xlnet_gpu.py -x -q -k "perm_mask or input_mask or deterministic" > $O/test_xl_new.txt 2>&1
timeout 600 python -m pytest tests/test_model_gpu.py -x -q -k "dropout_mask_replay" -s > $O/test_replay.txt 2>&1
timeout 900 python -m pytest tests -m gpu -q -x --deselect tests/test_dp_gpu.py::test_single_call_dp_step_two_ranks_equal_one_process > $O/test_all.txt 2>&1
MB_DP_FORCE=1 timeout 300 python bench.py --cpu-baseline 0 --roofline 0 --steps 100 --warmup 20 > $O/bench_dp_force.txt 2>&1
timeout 300 python bench.py --cpu-baseline 0 --roofline 0 --steps 100 --warmup 20 > $O/bench_plain.txt 2>&1
tail -3 $O/test_dp_new.txt $O/test_xl_new.txt $O/test_replay.txt $O/test_all.txt; cat $O/dp_step_bench.txt | cut -c1-220; tail -2 $O/bench_dp_force.txt | cut -c1-1500; tail -1 $O/bench_plain.txt | cut -c1-3000
