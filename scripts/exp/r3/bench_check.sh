#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
(timeout 400 python bench.py --cpu-steps 2 2>&1 | tail -3) > gpurun_out/r3_bench_check.txt 2>&1
(timeout 300 python bench.py --model xlnet --cpu-baseline 0 --steps 20 --warmup 5 2>&1 | tail -2) > gpurun_out/r3_bench_check_xlnet.txt 2>&1
cat /sys/fs/cgroup/cpu.max > gpurun_out/r3_cpu_max.txt 2>&1; nproc >> gpurun_out/r3_cpu_max.txt
tail -c 6000 gpurun_out/r3_bench_check.txt
