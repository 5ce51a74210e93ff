"""Reads a rocprofv3 kernel_trace.csv and reports, for the last `steps` training steps (delimited by adamw launches):
wall time, union-of-kernels busy time, idle time, per-queue busy time and the histogram of idle gaps."""
import csv
import sys

path, steps = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 5
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0")))
rows.sort()
is_ad = [("adamw" in r[2] and "tail" not in r[2]) for r in rows]
ends = [i for i in range(len(rows)) if is_ad[i] and (i + 1 == len(rows) or not is_ad[i + 1])]      # a step ends at the last launch of its AdamW run
lo, hi = ends[-steps - 1] + 1, ends[-1] + 1
seg = rows[lo:hi]
t0, t1 = seg[0][0], max(r[1] for r in seg)
busy, cur_s, cur_e, gaps = 0, seg[0][0], seg[0][1], []
for s, e, _, _ in seg[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append(s - cur_e)
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
wall = t1 - t0
print("steps %d kernels/step %.0f wall/step %.3f ms busy/step %.3f ms idle/step %.3f ms (%.1f%%)" %
      (steps, len(seg) / steps, wall / steps / 1e6, busy / steps / 1e6, (wall - busy) / steps / 1e6, 100.0 * (wall - busy) / wall))
ksum = sum(e - s for s, e, _, _ in seg)
print("sum of kernel durations / step %.3f ms (overlap factor %.2f)" % (ksum / steps / 1e6, ksum / busy))
q = {}
for s, e, _, qq in seg:
    q[qq] = q.get(qq, 0) + e - s
print("per-queue kernel time / step (ms):", {k: round(v / steps / 1e6, 3) for k, v in q.items()})
gaps.sort()
n = len(gaps)
if n:
    print("idle gaps/step %.0f  median %.2f us  p90 %.2f us  max %.1f us ; gaps > 5us: %d/step totalling %.3f ms/step" %
          (n / steps, gaps[n // 2] / 1e3, gaps[int(n * 0.9)] / 1e3, gaps[-1] / 1e3, sum(1 for g in gaps if g > 5000) / steps,
           sum(g for g in gaps if g > 5000) / steps / 1e6))
# biggest gaps: which kernel follows
big = []
prev_e = seg[0][1]
pe = seg[0][1]
for i in range(1, len(seg)):
    s, e, nme, _ = seg[i]
    if s - pe > 8000:
        big.append((s - pe, seg[i - 1][2][:50], nme[:50]))
    pe = max(pe, e)
big.sort(reverse=True)
for g, a, b in big[:12]:
    print("  gap %.1f us between %s -> %s" % (g / 1e3, a, b))

# per-kernel in-step durations (the same kernel is slower here than back-to-back: cold operands + CU sharing)
agg = {}
for s_, e_, nme, qq in seg:
    k = (nme[:64], qq)
    a = agg.setdefault(k, [0, 0])
    a[0] += 1; a[1] += e_ - s_
print("top kernels inside the step (name, queue): launches/step, avg us, ms/step")
for (nme, qq), (cnt, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:16]:
    print("  %-64s q%s %5.1f %8.1f %7.3f" % (nme, qq, cnt / steps, tot / cnt / 1e3, tot / steps / 1e6))
# wall time from the first backward kernel (head_bwd) to the last kernel before adamw
import re
