"""rocprofv3 kernel_trace.csv -> profiles/instep_kernels.json: per-kernel launches / step, average in-step duration, ms / step,
plus the step's wall / busy / idle time, over the last `steps` training steps (delimited by the AdamW launches).
usage: instep_json.py trace.csv steps "workload tag" out.json"""
import csv, json, sys
path, steps, tag, out = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4]
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
is_ad = [("adamw" in r[2] and "tail" not in r[2]) for r in rows]
ends = [i for i in range(len(rows)) if is_ad[i] and (i + 1 == len(rows) or not is_ad[i + 1])]      # a step ends at the last launch of its AdamW run
lo, hi = ends[-steps - 1] + 1, ends[-1] + 1
seg = rows[lo:hi]
t0, t1 = seg[0][0], max(r[1] for r in seg)
busy, cs, ce = 0, seg[0][0], seg[0][1]
for s, e, _ in seg[1:]:
    if s > ce:
        busy += ce - cs
        cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
agg = {}
for s, e, k in seg:
    name = k.split("(")[0]
    a = agg.setdefault(name, [0, 0])
    a[0] += 1; a[1] += e - s
ks = sorted(agg.items(), key=lambda kv: -kv[1][1])
doc = {"workload": tag, "source": "rocprofv3 --kernel-trace over tools/bin/step_bench (the C ABI step bench.py runs), last %d steps" % steps,
       "wall_ms_per_step": round((t1 - t0) / steps / 1e6, 4), "busy_ms_per_step": round(busy / steps / 1e6, 4),
       "traced_gap_frac": round(1.0 - busy / (t1 - t0), 4),
       "traced_gap_note": "gaps between kernels UNDER THE PROFILER (a traced graph replay serialises on the tool: wall here is 2-3x the "
                          "untraced step); not idle time of the real step -- compare busy_ms_per_step with the untraced ms_per_step",
       "kernels_per_step": round(len(seg) / steps, 1),
       "kernels": [{"kernel": k[:110], "launches_per_step": round(n / steps, 2), "avg_us": round(t / n / 1e3, 2),
                    "ms_per_step": round(t / steps / 1e6, 4)} for k, (n, t) in ks[:24]]}
json.dump(doc, open(out, "w"), indent=1)
print(json.dumps({k: doc[k] for k in ("wall_ms_per_step", "busy_ms_per_step", "traced_gap_frac", "kernels_per_step")}))
for k in doc["kernels"][:18]:
    print("%-100s %6.2f x %8.2f us = %.4f ms" % (k["kernel"][:100], k["launches_per_step"], k["avg_us"], k["ms_per_step"]))
