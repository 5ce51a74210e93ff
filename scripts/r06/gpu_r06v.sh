#!/bin/bash
mkdir -p gpurun_out/r06v
( time timeout 900 python bench.py ) > gpurun_out/r06v/bench.log 2>&1
grep '^{"metric' gpurun_out/r06v/bench.log > gpurun_out/r06v/bench_line.json
tail -5 gpurun_out/r06v/bench.log | cut -c1-300
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r06v/bench_line.json"))
keys = ["value", "ms_per_step", "value_with_per_step_h2d", "step_ms_median", "epoch_ms", "epoch_train_ms", "epoch_eval_ms", "epoch_test_ms", "epoch_train_outside_full_steps_ms", "epoch_error", "gpu_over_cpu"]
print({k: d.get(k) for k in keys})
r = d["roofline"]; print("roofline:", {k: r.get(k) for k in ("kernel", "bound", "achieved", "frac", "avg_us", "ms_per_step", "traffic")})
for x in d.get("roofline_trace", [])[:6]: print("  ", x["kernel"][:70], x["ms_per_step"], x["frac"], x.get("parameters_swept_per_step"))
print("gemm_aggregate", d.get("instep_kernels", {}).get("gemm_aggregate"), d.get("instep_kernels", {}).get("adamw_riders"))
for s in d.get("secondary", []):
    print("secondary:", s.get("metric"), s.get("value"), s.get("ms_per_step"), s.get("value_with_per_step_h2d"), s.get("error"))
    r = s.get("roofline") or {}
    print("    roofline:", {k: r.get(k) for k in ("kernel", "bound", "achieved", "frac", "ms_per_step")}, s.get("roofline_note"))
    print("    gemm_aggregate:", s.get("gemm_aggregate"), "cpu:", (s.get("cpu_baseline") or {}).get("value"))
print("cpu_baseline", d.get("cpu_baseline", {}).get("value"), d.get("cpu_baseline", {}).get("cores"))
PY
