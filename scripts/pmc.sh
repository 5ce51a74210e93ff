#!/bin/bash
# rocprofv3 PMC passes over the GEMM micro-benchmark (separate passes, kernel-trace only -- no sys/hip trace domains)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/pmc
rocprofv3 -L > $R/gpurun_out/pmc/counters.txt 2>&1
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES" \
           "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE TCP_TCC_READ_REQ_sum" "GRBM_GUI_ACTIVE TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  rm -rf /tmp/pmc$i
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmc$i -o p -- python $R/bench.py --roofline-only 1 > /tmp/pmc$i.log 2>&1
  f=$(find /tmp/pmc$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp $f $R/gpurun_out/pmc/pass$i.csv || tail -5 /tmp/pmc$i.log
done
ls -la $R/gpurun_out/pmc
