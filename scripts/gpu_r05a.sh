#!/bin/bash
# round 5: the data-parallel step on ONE GPU after the hand-off fix (event nodes that really order the replay), same box, twice each:
# single call | --dp 1 (one-rank RCCL communicator) in event modes 0 / 2 / 3 | --dp 1 --shard 1 (MB_DP_SHARD_FORCE=1: the sharded
# update's code path with identity collectives).   usage: bash scripts/gpu_r05a.sh  -> gpurun_out/r05/dp_event_modes.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05; mkdir -p $O; cd $R
SB=$R/tools/bin/step_bench
export HSA_ENABLE_IPC_MODE_LEGACY=0
{
for rep in 1 2; do
  echo "== single call (mb_bert_train_step)"; timeout 120 $SB --graph 1 --h2d 2 --steps 200 --warmup 20 | grep ms/step
  for mode in 0 2 3; do
    echo "== --dp 1, MB_DP_EVENT_MODE=$mode"; MB_DP_EVENT_MODE=$mode timeout 180 $SB --graph 1 --h2d 2 --steps 200 --warmup 20 --dp 1 2>&1 | grep "ms/step"
  done
  echo "== --dp 1 --shard 1 (MB_DP_SHARD_FORCE=1), mode 2"; MB_DP_SHARD_FORCE=1 timeout 180 $SB --graph 1 --h2d 2 --steps 200 --warmup 20 --dp 1 --shard 1 2>&1 | grep "ms/step\|step_bench dp"
  echo "== --dp 1 --shard 1 (MB_DP_SHARD_FORCE=1), mode 3"; MB_DP_EVENT_MODE=3 MB_DP_SHARD_FORCE=1 timeout 180 $SB --graph 1 --h2d 2 --steps 200 --warmup 20 --dp 1 --shard 1 2>&1 | grep "ms/step"
done
for chunks in "6,6" "12" "4,4,4" "2,2,2,2,2,2"; do
  echo "== --dp 1, mode 2, MB_DP_CHUNKS=$chunks (layers per backward segment)"; MB_DP_CHUNKS=$chunks timeout 180 $SB --graph 1 --h2d 2 --steps 200 --warmup 20 --dp 1 2>&1 | grep "ms/step"
done
} > $O/dp_event_modes.txt 2>&1
cat $O/dp_event_modes.txt
