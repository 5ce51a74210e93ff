"""pmc_FETCH_SIZE.txt / pmc_WRITE_SIZE.txt (scripts/exp/pmc_reduce.py: per-kernel means of the rocprofv3 counters, unit KB) ->
profiles/pmc_traffic.json, the HBM-side bytes bench.py replays next to its live roofline numbers.
usage: pmc_traffic_json.py <dir with pmc_*.txt> "workload tag" out.json
FETCH_SIZE is doubled (MI355X_MICROARCH.md: gfx950 reports half the bytes of wide coalesced reads); WRITE_SIZE as is."""
import json, sys
d, tag, out = sys.argv[1], sys.argv[2], sys.argv[3]


def table(path):
    t = {}
    for line in open(path):
        if line.startswith("#") or not line.strip():
            continue
        f = line.split()
        t[" ".join(f[:-3])] = (int(f[-3]), float(f[-2]))      # kernel -> (launches, mean per launch in KB)
    return t


fe, wr = table(d + "/pmc_FETCH_SIZE.txt"), table(d + "/pmc_WRITE_SIZE.txt")
pick = lambda t, key: next(((n, v) for k, (n, v) in t.items() if key in k), (0, 0.0))
doc = {"workload": tag, "kernels": {},
       "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over tools/bin/step_bench (separate passes, kernels launched one by one), "
                 "FETCH x2 on gfx950 per MI355X_MICROARCH.md, calibrated on the optimizer's known read volume"}
nf, f = pick(fe, "grouped_tn_kernelIDF16bLi128"); nw, w = pick(wr, "grouped_tn_kernelIDF16bLi128")
doc["kernels"]["wgrad_grouped"] = {"fetch_bytes": int(2 * f * 1024), "write_bytes": int(w * 1024), "algorithmic_bytes": 88080384,
                                   "note": "per launch; algorithmic = dY 33.6 MB + X 26.1 MB read + dW 28.3 MB stored (known-zero gradients)"}
nf, f = pick(fe, "adamw"); nw, w = pick(wr, "adamw")
# two launches per step (decay / no-decay group): the table holds the mean over both, so x2 = bytes per step
doc["kernels"]["adamw"] = {"fetch_bytes": int(2 * f * 1024 * 2), "write_bytes": int(w * 1024 * 2), "algorithmic_bytes": 28 * 110853121,
                           "note": "per step (both launches); reads p, g, m, v = 16 B/param = 1.774 GB: the calibration point of the x2 correction"}
# every symbol both passes saw, keyed by the first 90 characters of its name (what pmc_reduce.py keeps): bench.py looks its
# in-run trace's dominant symbol up here, whichever kernel that is
doc["by_symbol"] = {k: {"launches": n, "fetch_bytes": int(2 * v * 1024), "write_bytes": int(wr[k][1] * 1024)}
                    for k, (n, v) in fe.items() if k in wr and "adamw" not in k}
json.dump(doc, open(out, "w"), indent=1)
print(json.dumps(doc["kernels"]))
