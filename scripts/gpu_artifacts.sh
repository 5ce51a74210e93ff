#!/bin/bash
# round artefacts: bench lines (headline, C5 shape, MAG-XLNet), rocprofv3 kernel stats, in-step kernel table, PMC passes.
# usage: bash scripts/gpu_artifacts.sh r04     (every command under its own timeout)
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-r04}
O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
SB=$R/tools/bin/step_bench
export TMPDIR=/tmp
# ---- 1. in-step kernel table + idle, graph replay (what bench.py runs) and PMC passes, torch-free
( cd /tmp && rm -rf /tmp/p_tr && timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_tr -o sb -- $SB --graph 1 --h2d 2 --steps 20 --warmup 5 2>&1 | grep step_bench ) > $O/step_bench_traced.txt
f=$(find /tmp/p_tr -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python3 scripts/exp/instep_json.py $f 8 "bert B=48 L=50 bf16" $O/instep_kernels.json > $O/instep_kernels.txt
f=$(find /tmp/p_tr -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/step_kernel_stats.csv
for set in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && rm -rf /tmp/p_$set && MB_GEMM_LOG=1 timeout 120 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/p_$set -o p -- $SB --graph 2 --h2d 2 --steps 6 --warmup 2 2>&1 > /dev/null | grep "magbert adamw" | tail -n 8 > $O/pmc_adamw_log.txt )
  f=$(find /tmp/p_$set -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python3 scripts/exp/pmc_reduce.py $f $set > $O/pmc_$set.txt
done
python3 scripts/exp/pmc_traffic_json.py $O "bert B=48 L=50 bf16" $O/pmc_traffic.json > $O/pmc_traffic.txt 2>&1 && cp $O/pmc_traffic.json profiles/pmc_traffic.json
( cd /tmp && rm -rf /tmp/p_sq && timeout 120 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES --output-format csv -d /tmp/p_sq -o p -- $SB --graph 2 --h2d 2 --steps 6 --warmup 2 > /dev/null 2>&1 )
f=$(find /tmp/p_sq -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python3 - "$f" > $O/pmc_SQ.txt <<'PY'
import csv, sys
agg = {}
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        d = agg.setdefault(r["Kernel_Name"][:90], {})
        a = d.setdefault(r["Counter_Name"], [0, 0.0]); a[0] += 1; a[1] += float(r["Counter_Value"])
rows = []
for k, d in agg.items():
    g = lambda n: d.get(n, [1, 0.0])[1] / max(1, d.get(n, [1, 0.0])[0])
    wc = g("SQ_WAVE_CYCLES")
    if wc > 0:
        rows.append((d["SQ_WAVE_CYCLES"][1], k, d["SQ_WAVE_CYCLES"][0], g("SQ_WAVES"), wc, g("SQ_VALU_MFMA_BUSY_CYCLES"), g("SQ_INSTS_VALU_MFMA_MOPS_BF16"),
                     g("SQ_WAIT_ANY") / wc, g("SQ_WAIT_INST_ANY") / wc, g("SQ_ACTIVE_INST_ANY") / wc))
print("# per launch: kernel, launches, waves, wave quad-cycles, MFMA busy cycles, MFMA MOPS(bf16), wait_any/wave_cyc, wait_inst/wave_cyc, active_inst/wave_cyc")
for _, k, n, w, wc, mb, mo, wa, wi, ai in sorted(rows, reverse=True)[:14]:
    print("%-90s %5d %7.0f %12.0f %12.0f %12.0f %5.2f %5.2f %5.2f" % (k, n, w, wc, mb, mo, wa, wi, ai))
PY
timeout 60 tools/bin/gemm_bench > $O/gemm_bench.txt 2>&1
timeout 60 tools/bin/gemm_bench --T 4096 > $O/gemm_bench_T4096.txt 2>&1
# per-block phase stamps (100 MHz wall clock) of the GEMM launches and of the attention backward; launch floor
MB_GEMM_TRACE=1 timeout 60 tools/bin/gemm_bench --trace 1 > $O/gemm_phases.txt 2>&1
{ MB_ATTN_TRACE=1 timeout 60 tools/bin/attn_bench; MB_ATTN_TRACE=1 timeout 60 tools/bin/attn_bench --batch 32 --seq 128; timeout 60 tools/bin/attn_bench; } > $O/attention_phases.txt 2>&1
timeout 60 tools/bin/launch_floor > $O/launch_floor.txt 2>&1
# per-iteration stamps inside the k loop (library built with -DMB_GEMM_LOOPTRACE by scripts/build_variant.py, if present)
[ -f gpurun_ab/looptrace/libmagbert_hip.so ] && LD_LIBRARY_PATH=$R/gpurun_ab/looptrace:$LD_LIBRARY_PATH MB_GEMM_TRACE=1 timeout 120 tools/bin/gemm_bench --looptrace 1 > $O/gemm_looptrace.txt 2>&1
# same-box A/B of the software-pipelined k loop against the plain loop (-DMB_GEMM_PLAIN_LOOP build, if present)
[ -f gpurun_ab/plainloop/libmagbert_hip.so ] && REPS=2 bash scripts/gpu_ab.sh plainloop > $O/ab_pipelined_vs_plain.txt 2>&1
{ timeout 60 tools/bin/adamw_bench; MB_ADAMW_VAR=0 timeout 60 tools/bin/adamw_bench; timeout 60 tools/bin/adamw_bench --zero 0; } > $O/adamw_bench.txt 2>&1
# clocks / power while the step runs (sustained load: what the in-step kernel durations are measured at)
{ timeout 60 $SB --graph 1 --h2d 2 --steps 3000 --warmup 10 > $O/clocks_step.txt 2>&1 & SBPID=$!
  sleep 4; for k in 1 2 3; do timeout 10 rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|mclk|fclk|Power|power" ; echo --; sleep 2; done; wait $SBPID; cat $O/clocks_step.txt; } > $O/clocks_under_load.txt 2>&1
# ---- 2. C5 shape and a second headline timing from the C++ driver
{ timeout 60 $SB --graph 1 --h2d 2 --steps 40 --warmup 8; timeout 60 $SB --graph 1 --h2d 2 --batch 32 --seq 128 --visual 35 --steps 30 --warmup 6; } > $O/step_bench.txt 2>&1
( cd /tmp && rm -rf /tmp/p_c5 && timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c5 -o sb -- $SB --graph 1 --h2d 2 --batch 32 --seq 128 --visual 35 --steps 12 --warmup 4 > /dev/null 2>&1 )
f=$(find /tmp/p_c5 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/c5_kernel_stats.csv
f=$(find /tmp/p_c5 -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python3 scripts/exp/instep_json.py $f 6 "bert B=32 L=128 bf16 (MOSEI V=35)" $O/instep_kernels_c5.json > $O/instep_kernels_c5.txt
# ---- 3. bench.py lines
cp $O/instep_kernels.json profiles/instep_kernels.json 2>/dev/null
J='^{"metric'
(timeout 500 python bench.py 2>&1 | grep "$J") > $O/bench_line.log 2>&1
(timeout 300 python bench.py --dataset mosei --seq 128 --batch 32 --cpu-baseline 0 --steps 30 --warmup 6 2>&1 | grep "$J") > $O/bench_line_c5.log 2>&1
(timeout 300 python bench.py --model xlnet --cpu-steps 2 --steps 30 --warmup 6 2>&1 | grep "$J") > $O/bench_line_xlnet.log 2>&1
# ---- 3b. the data-parallel step on ONE GPU (one-rank RCCL group, MB_DP_FORCE=1) against the single-call step, same box, twice each
export HSA_ENABLE_IPC_MODE_LEGACY=0
{
for rep in 1 2; do
  echo "== single call (mb_bert_train_step), bench.py"; timeout 300 python bench.py --cpu-baseline 0 --roofline 0 --secondary 0 --steps 100 --warmup 20 2>&1 | grep "$J" | cut -c1-1400
  echo "== MB_DP_FORCE=1: mb_bert_train_step_dp, one-rank RCCL communicator driven from C, bench.py"; MB_DP_FORCE=1 timeout 300 python bench.py --cpu-baseline 0 --roofline 0 --steps 100 --warmup 20 2>&1 | grep "$J" | cut -c1-1700
done
echo "== MB_DP_FORCE=1 MB_DP_ENGINE=0: round-3 structure (passes and exchange driven from Python)"; MB_DP_FORCE=1 MB_DP_ENGINE=0 timeout 300 python bench.py --cpu-baseline 0 --roofline 0 --steps 100 --warmup 20 2>&1 | grep "$J" | cut -c1-1500
echo "== MB_DP_FORCE=1 MB_DP_GRAD_DTYPE=bf16: the two-GPU wire format forced on one GPU (staging passes)"; MB_DP_FORCE=1 MB_DP_GRAD_DTYPE=bf16 timeout 300 python bench.py --cpu-baseline 0 --roofline 0 --steps 100 --warmup 20 2>&1 | grep "$J" | cut -c1-1500
echo "== MAG-XLNet single call / MB_DP_FORCE=1"; timeout 300 python bench.py --model xlnet --cpu-baseline 0 --roofline 0 --steps 60 --warmup 10 2>&1 | grep "$J" | cut -c1-1200
MB_DP_FORCE=1 timeout 300 python bench.py --model xlnet --cpu-baseline 0 --roofline 0 --steps 60 --warmup 10 2>&1 | grep "$J" | cut -c1-1500
for rep in 1 2; do
  echo "== C++ driver: single call"; timeout 120 $SB --graph 1 --h2d 2 --steps 100 --warmup 20
  echo "== C++ driver: --dp 1 (one-rank RCCL communicator)"; timeout 180 $SB --graph 1 --h2d 2 --steps 100 --warmup 20 --dp 1 2>&1 | grep step_bench
  echo "== C++ driver: --dp 1, host cost with an empty queue (5 steps)"; timeout 180 $SB --graph 1 --h2d 2 --steps 5 --warmup 20 --dp 1 2>&1 | grep "ms/step"
done
} > $O/dp_force.txt 2>&1
( cd /tmp && rm -rf /tmp/p_b && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_b -o b -- python $R/bench.py --steps 10 --warmup 3 --cpu-baseline 0 --roofline 0 --secondary 0 > /dev/null 2>&1 )
f=$(find /tmp/p_b -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/bench_kernel_stats.csv
( cd /tmp && rm -rf /tmp/p_x && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_x -o b -- python $R/bench.py --model xlnet --steps 10 --warmup 3 --cpu-baseline 0 --roofline 0 > /dev/null 2>&1 )
f=$(find /tmp/p_x -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/xlnet_kernel_stats.csv
# bench.py starting its own ranks (round 5): two ranks sharing this GPU over the callback backend -- the launcher, the rccl_ranks proof and the JSON contract, not a speed
(MB_DIST_BACKEND=gloo timeout 300 python bench.py --gpus 2 --cpu-baseline 0 --roofline 0 --steps 10 --warmup 3 2>&1 | grep "$J" | cut -c1-1800) > $O/bench_line_gpus2_gloo.log 2>&1
{ timeout 60 python bench.py --gpus 2 --cpu-baseline 0 --roofline 0 > $O/.g2.tmp 2>&1; rc=$?; tail -n 2 $O/.g2.tmp; rm -f $O/.g2.tmp; echo "exit code $rc"; } > $O/bench_gpus2_without_devices.log 2>&1
for f in bench_line bench_line_c5 bench_line_xlnet; do tail -n 1 $O/$f.log | cut -c1-420; done
cat $O/instep_kernels.txt | head -24
cat $O/step_bench.txt
