#!/bin/bash
# round 6: data-parallel seams on ONE GPU, one-rank RCCL group, torch-free driver.
#   part 1: MB_DP_CHUNKS = 4,4,2,2 | 6,6 | 12 | 4,4,4 | 2x6, without and with a synthetic link load behind every all-reduce piece
#           (MB_DP_SYNTH_GBPS=300: the piece copied onto itself by 8 workgroups for bytes / 300 GB/s)
#   part 2: the comm-stream priority inside a PyTorch process under GPU_MAX_HW_QUEUES (bench.py, MB_DP_FORCE=1)
mkdir -p gpurun_out/r06x
O=gpurun_out/r06x/dp_chunks.txt
: > $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
SB=tools/bin/step_bench
ARGS="--graph 1 --h2d 2 --steps 150 --warmup 20"
echo "== single call (no data parallel)" >> $O
timeout 120 $SB $ARGS 2>&1 | grep "ms/step" | cut -c60-140 >> $O
for synth in 0 300 150; do
  for ch in "4,4,2,2" "6,6" "12" "4,4,4" "2,2,2,2,2,2" "8,4"; do
    echo "== MB_DP_SYNTH_GBPS=$synth MB_DP_CHUNKS=$ch" >> $O
    MB_DP_SYNTH_GBPS=$synth MB_DP_CHUNKS=$ch timeout 180 $SB $ARGS --dp 1 --timing 1 2>&1 | grep "ms/step\|comm_exposed" | cut -c1-200 | sed 's/step_bench dtype=bf16 B=48 L=50 V=47 layers=12 graph=1 h2d=2 ://' >> $O
  done
done
echo "== part 2: bench.py MB_DP_FORCE=1 (PyTorch process), comm stream priority x GPU_MAX_HW_QUEUES" >> $O
J='^{"metric'
for prio in 0 1; do
  for q in default 2 4 8; do
    echo "== MB_DP_COMM_PRIORITY=$prio GPU_MAX_HW_QUEUES=$q" >> $O
    if [ "$q" = default ]; then MB_DP_FORCE=1 MB_DP_COMM_PRIORITY=$prio timeout 300 python bench.py --cpu-baseline 0 --roofline 0 --secondary 0 --epoch 0 --steps 60 --warmup 10 2>&1 | grep "$J" | python3 -c "import sys, json; d = json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('comm_exposed_ms'), d.get('host_call_ms_per_step'))" >> $O
    else GPU_MAX_HW_QUEUES=$q MB_DP_FORCE=1 MB_DP_COMM_PRIORITY=$prio timeout 300 python bench.py --cpu-baseline 0 --roofline 0 --secondary 0 --epoch 0 --steps 60 --warmup 10 2>&1 | grep "$J" | python3 -c "import sys, json; d = json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('comm_exposed_ms'), d.get('host_call_ms_per_step'))" >> $O; fi
  done
done
cat $O
