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
# the layers' grouped weight gradient: the ping-pong kernel (round 6), else the 128 x 128 grouped kernel
key = "gemm_pp_grouped_tn_kernel" if any("gemm_pp_grouped_tn_kernel" in k for k in fe) else "grouped_tn_kernelIDF16bLi128"
nf, f = pick(fe, key); nw, w = pick(wr, key)
doc["kernels"]["wgrad_grouped"] = {"fetch_bytes": int(2 * f * 1024), "write_bytes": int(w * 1024), "algorithmic_bytes": 88080384,
                                   "note": "per launch (mean over the 12 layers' launches); algorithmic = dY 33.6 MB + X 26.1 MB read + dW 28.3 MB stored "
                                           "(known-zero gradients).  Since round 6 eleven of the twelve launches also carry AdamW riders "
                                           "(csrc/kernels.h AdamRide: ~2.43 M parameters each = 39 MB read + 44 MB written by design), included here"}
steps = max(1, nf // 12)                      # 12 layers -> one grouped launch per layer and step
na, f = pick(fe, "adamw"); nw, w = pick(wr, "adamw")
lps = max(1, round(na / steps))               # sweep launches per step: 2 without riders (decay / no-decay), 3 with (two decay ranges around the ridden one)
# parameters the sweep launches of one step cover: the library's own log of the pass (MB_GEMM_LOG=1: one "[magbert adamw] n=" line per launch)
swept = None
try:
    ns = [int(l.split("=")[1]) for l in open(d + "/pmc_adamw_log.txt") if l.startswith("[magbert adamw] n=")]
    if len(ns) >= lps:
        swept = sum(ns[-lps:])
except OSError:
    pass
ridden = (110853121 - swept) if swept else (11 * 2432000 if lps == 3 else 0)
doc["kernels"]["adamw"] = {"fetch_bytes": int(2 * f * 1024 * lps), "write_bytes": int(w * 1024 * lps), "launches_per_step": lps,
                           "algorithmic_bytes": 28 * (110853121 - ridden),
                           "note": "per step (the %d sweep launches%s); reads p, g, m, v = 16 B/param: the calibration point of the x2 correction"
                                   % (lps, "; %.1f M of the 110.9 M parameters are updated by riders inside the weight-gradient, dgrad and attention-backward launches instead" % (ridden / 1e6) if ridden else "")}
# every symbol both passes saw, keyed by the first 90 characters of its name (what pmc_reduce.py keeps): bench.py looks its
# in-run trace's dominant symbol up here, whichever kernel that is
doc["by_symbol"] = {k: {"launches": n, "fetch_bytes": int(2 * v * 1024), "write_bytes": int(wr[k][1] * 1024)}
                    for k, (n, v) in fe.items() if k in wr and "adamw" not in k}
json.dump(doc, open(out, "w"), indent=1)
print(json.dumps(doc["kernels"]))
