#!/bin/bash
# One rocprofv3 PMC pass (SQ block only, --kernel-trace only) over the training step: MFMA busy / wait / issue shares per kernel.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/pmc_step
rm -rf /tmp/pmcsq
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES --output-format csv -d /tmp/pmcsq -o p -- python $R/bench.py --steps 6 --warmup 2 --cpu-baseline 0 --roofline 0 > /tmp/pmcsq.log 2>&1
f=$(find /tmp/pmcsq -name "*counter_collection.csv" | head -1)
if [ -z "$f" ]; then tail -5 /tmp/pmcsq.log; exit 1; fi
python - "$f" > $R/gpurun_out/pmc_step/SQ.txt <<'PY'
import csv, sys
agg = {}
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        k = r["Kernel_Name"][:70]
        d = agg.setdefault(k, {})
        a = d.setdefault(r["Counter_Name"], [0, 0.0])
        a[0] += 1; a[1] += float(r["Counter_Value"])
rows = []
for k, d in agg.items():
    g = lambda n: d.get(n, [1, 0.0])[1] / max(1, d.get(n, [1, 0.0])[0])
    wc = g("SQ_WAVE_CYCLES")
    if wc <= 0: continue
    rows.append((d["SQ_WAVE_CYCLES"][1], k, d["SQ_WAVE_CYCLES"][0], g("SQ_WAVES"), wc, g("SQ_VALU_MFMA_BUSY_CYCLES"), g("SQ_INSTS_VALU_MFMA_MOPS_BF16"),
                 g("SQ_WAIT_ANY") / wc, g("SQ_WAIT_INST_ANY") / wc, g("SQ_ACTIVE_INST_ANY") / wc))
rows.sort(reverse=True)
print("# per launch: kernel, launches, waves, wave quad-cycles, MFMA busy cycles, MFMA MOPS(bf16), wait_any/wave_cyc, wait_inst/wave_cyc, active_inst/wave_cyc")
for _, k, n, w, wc, mb, mo, wa, wi, ai in rows[:14]:
    print("%-70s %5d %7.0f %12.0f %12.0f %12.0f %5.2f %5.2f %5.2f" % (k, n, w, wc, mb, mo, wa, wi, ai))
PY
cat $R/gpurun_out/pmc_step/SQ.txt
