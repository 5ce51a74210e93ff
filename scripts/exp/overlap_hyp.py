import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_model_gpu import build, tb, weights, DEV
from bert_multimodal_transformer_amd import AdamW, get_linear_schedule_with_warmup
from bert_multimodal_transformer_amd.multimodal_driver import optimizer_grouped_parameters

def run(stale_stages=None, stale_what="params"):
    torch.manual_seed(3)
    m = build(layers=3, cdt=torch.bfloat16).train()
    core = m._core
    opt = AdamW(optimizer_grouped_parameters(m), lr=1e-4)
    sch = get_linear_schedule_with_warmup(opt, 0, 100)
    P0 = m.flat_params.clone(); S0 = core.shadow.clone()
    out = []
    for s in range(2):
        ids, vis, aco, mask, seg, lab = tb(weights.synthetic_bert_batch(8, 50, 47, 74, seed=70 + s), DEV)
        if s == 1 and stale_stages is not None:
            P1 = m.flat_params.clone(); S1 = core.shadow.clone()
            for st in stale_stages:
                for off, n in ([(st[1], st[2])] if isinstance(st, tuple) else core.stage_ranges(st)):
                    m.flat_params[off:off + n] = P0[off:off + n]
                    core.shadow[off:off + n] = S0[off:off + n]
            torch.cuda.synchronize()
        m.training_step(ids, vis, aco, mask, seg, lab)
        torch.cuda.synchronize()
        if s == 1 and stale_stages is not None:
            m.flat_params.copy_(P1); core.shadow.copy_(S1)
            torch.cuda.synchronize()
        opt.step(); sch.step(); opt.zero_grad()
        torch.cuda.synchronize()
        out.append(m.flat_params.clone())
    return out, core

ref, core = run()
table = core.tensors
q = [t for t in table if t[0] == "bert.encoder.layer.0.attention.self.query.weight"][0]
import types
def run_tensor(off, n):
    # monkeypatch: stale exactly [off, off+n)
    return run(stale_stages=[("range", off, n)])

rs = core.stage_ranges(4)
print("last-stage ranges:", rs)
for name, sel in (("last small A", [rs[1]]), ("last small B", [rs[2]]), ("last small A+B", [rs[1], rs[2]]), ("last big", [rs[0]])):
    o, _ = run(stale_stages=[("range", a, n) for a, n in sel])
    d = (o[1] - ref[1]).abs()
    print("%-24s max %.10e  layer0.query max %.10e frac %.6f" % (name, float(d.max()), float(d[q[1]:q[1] + q[2]].max()),
                                                                  float((d[q[1]:q[1] + q[2]] > 2e-6).float().mean())))
rs0 = core.stage_ranges(0)
for k, r in enumerate(rs0):
    o, _ = run(stale_stages=[("range", r[0], r[1])])
    d = (o[1] - ref[1]).abs()
    print("stage0 range %d %s max %.10e  layer0.query max %.10e frac %.6f" % (k, r, float(d.max()), float(d[q[1]:q[1] + q[2]].max()),
                                                                  float((d[q[1]:q[1] + q[2]] > 2e-6).float().mean())))
print("target B: max 1.2603463256e-04  layer0.query 4.6234577894e-05 frac 0.015216")
