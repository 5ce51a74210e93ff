#!/bin/bash
# round 5: the data-parallel step inside a PyTorch process after BOTH fixes (hand-off events that order the streams; comm stream at
# normal priority), one GPU, one-rank RCCL communicator (MB_DP_FORCE=1), same box, twice each.  -> gpurun_out/r05/dp_force_python.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05; mkdir -p $O; cd $R
SB=$R/tools/bin/step_bench
export HSA_ENABLE_IPC_MODE_LEGACY=0
J='^{"metric'
P='import sys,json; d=json.loads(sys.stdin.read()); print("   %.0f samples/s  %.3f ms/step  (median %.3f)  host call %.3f ms  comm_exposed %s ms  %s" % (d["value"], d["ms_per_step"], d["step_ms_median"], d["host_call_ms_per_step"], d.get("comm_exposed_ms"), d["config"].get("step_call", "")[:60]))'
B="--cpu-baseline 0 --roofline 0 --secondary 0 --steps 100 --warmup 20"
{
for rep in 1 2; do
  echo "== bench.py, single call (mb_bert_train_step)"; timeout 300 python bench.py $B 2>&1 | grep "$J" | python -c "$P"
  echo "== bench.py, MB_DP_FORCE=1: mb_bert_train_step_dp (event-record nodes, comm stream at normal priority)"; MB_DP_FORCE=1 timeout 300 python bench.py $B 2>&1 | grep "$J" | python -c "$P"
done
echo "== bench.py, MB_DP_FORCE=1 MB_DP_COMM_PRIORITY=1 (round 4's highest-priority comm stream)"; MB_DP_FORCE=1 MB_DP_COMM_PRIORITY=1 timeout 300 python bench.py $B 2>&1 | grep "$J" | python -c "$P"
echo "== bench.py, MB_DP_FORCE=1 MB_DP_SHARD_OPT=1 MB_DP_SHARD_FORCE=1 (sharded update's code path, identity collectives)"; MB_DP_FORCE=1 MB_DP_SHARD_OPT=1 MB_DP_SHARD_FORCE=1 timeout 300 python bench.py $B 2>&1 | grep "$J" | python -c "$P"
echo "== bench.py, MB_DP_FORCE=1 MB_DP_ENGINE=0 (round-3 structure: passes and exchange driven from Python)"; MB_DP_FORCE=1 MB_DP_ENGINE=0 timeout 300 python bench.py $B 2>&1 | grep "$J" | python -c "$P"
echo "== bench.py, MB_DP_FORCE=1 MB_DP_GRAD_DTYPE=bf16 (the two-GPU wire format forced on one GPU)"; MB_DP_FORCE=1 MB_DP_GRAD_DTYPE=bf16 timeout 300 python bench.py $B 2>&1 | grep "$J" | python -c "$P"
echo "== bench.py --model xlnet, single call / MB_DP_FORCE=1"; timeout 300 python bench.py --model xlnet $B 2>&1 | grep "$J" | python -c "$P"; MB_DP_FORCE=1 timeout 300 python bench.py --model xlnet $B 2>&1 | grep "$J" | python -c "$P"
for rep in 1 2; do
  echo "== C++ driver: single call"; timeout 120 $SB --graph 1 --h2d 2 --steps 200 --warmup 20 | grep ms/step
  echo "== C++ driver: --dp 1 (normal priority)"; timeout 180 $SB --graph 1 --h2d 2 --steps 200 --warmup 20 --dp 1 2>&1 | grep "ms/step"
  echo "== C++ driver: --dp 1, MB_DP_COMM_PRIORITY=1"; MB_DP_COMM_PRIORITY=1 timeout 180 $SB --graph 1 --h2d 2 --steps 200 --warmup 20 --dp 1 2>&1 | grep "ms/step"
done
} > $O/dp_force_python.txt 2>&1
cat $O/dp_force_python.txt
