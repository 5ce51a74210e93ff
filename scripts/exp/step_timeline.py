"""Position-by-position timeline of one training step: a rocprofv3 kernel_trace.csv of tools/bin/step_bench (steps end at the last launch of
every AdamW run) averaged over the last `steps` steps, next to the MB_GEMM_LOG=1 lines of the same run (one per GEMM launch of the
enqueue pass) -- what each call site costs IN the step (cold operands), which the per-symbol tables average away.
usage: step_timeline.py kernel_trace.csv gemm_log.txt [steps]"""
import csv
import re
import sys

path, logp, steps = sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 10
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
is_ad = [("adamw" in x[2] and "tail" not in x[2]) for x in rows]
ends = [i for i in range(len(rows)) if is_ad[i] and (i + 1 == len(rows) or not is_ad[i + 1])]
segs = [rows[ends[k] + 1: ends[k + 1] + 1] for k in range(len(ends) - 1)][-steps:]
n = len(segs[-1])
segs = [s for s in segs if len(s) == n and all(a[2] == b[2] for a, b in zip(s, segs[-1]))]
gl = [l.split() for l in open(logp) if l.startswith("[magbert gemm] ")]


def short(k):
    m = re.match(r"_ZN2mb\d+([a-z0-9_]+?)I", k)
    if m:
        t = re.findall(r"Li(\d+)E|Lb([01])E", k)
        return m.group(1) + "<" + ",".join(a or b for a, b in t[:8]) + ">"
    return k.split("(")[0].replace("void ", "")[:48]


# the enqueue pass logs every GEMM launch once, in launch order: the last len(gemm kernels in a step) lines belong to the captured step
gk = [i for i, r in enumerate(segs[-1]) if "gemm" in r[2]]
lines = gl[-len(gk):] if len(gl) >= len(gk) else []
shape = {}
for i, w in zip(gk, lines):
    kv = dict(t.split("=") for t in w[3:] if "=" in t)
    shape[i] = "M=%s N=%s K=%s" % (kv.get("M"), kv.get("N"), kv.get("K"))
print("# %d kernels per step, averaged over %d steps; columns: position, start offset us, duration us, gap before us, kernel, shape" % (n, len(segs)))
tot = 0.0
for i in range(n):
    d = sum(s[i][1] - s[i][0] for s in segs) / len(segs) / 1e3
    st = sum(s[i][0] - s[0][0] for s in segs) / len(segs) / 1e3
    gp = sum((s[i][0] - s[i - 1][1]) if i else 0 for s in segs) / len(segs) / 1e3
    tot += d
    print("%3d %9.1f %7.2f %6.2f  %-56s %s" % (i, st, d, gp, short(segs[-1][i][2]), shape.get(i, "")))
print("# sum of durations %.1f us ; first start to last end %.1f us" % (tot, sum(s[-1][1] - s[0][0] for s in segs) / len(segs) / 1e3))
