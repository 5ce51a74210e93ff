"""per-kernel mean of one rocprofv3 counter from a counter_collection.csv"""
import csv, sys
path, name = sys.argv[1], sys.argv[2]
agg = {}
with open(path) as f:
    for r in csv.DictReader(f):
        if r.get("Counter_Name") != name:
            continue
        a = agg.setdefault(r["Kernel_Name"][:90], [0, 0.0])
        a[0] += 1; a[1] += float(r["Counter_Value"])
print("# %s: kernel, launches, mean per launch, total" % name)
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
    print("%-90s %6d %14.1f %16.1f" % (k, n, t / n, t))
