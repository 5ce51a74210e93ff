import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from test_model_gpu import build, tb, weights, DEV
from bert_multimodal_transformer_amd import AdamW, get_linear_schedule_with_warmup
from bert_multimodal_transformer_amd.multimodal_driver import optimizer_grouped_parameters

def run(overlap, ref=None, steps=3):
    torch.manual_seed(3)
    m = build(layers=3, cdt=torch.bfloat16).train()
    opt = AdamW(optimizer_grouped_parameters(m), lr=1e-4)
    sch = get_linear_schedule_with_warmup(opt, 0, 100)
    if overlap:
        opt.enable_overlap(m)
    snaps = []
    for s in range(steps):
        ids, vis, aco, mask, seg, lab = tb(weights.synthetic_bert_batch(8, 50, 47, 74, seed=70 + s), DEV)
        m.training_step(ids, vis, aco, mask, seg, lab)
        opt.step(); sch.step(); opt.zero_grad()
        if os.environ.get("DIAG_SYNC"):
            torch.cuda.synchronize()
        if os.environ.get("DIAG_FINAL_ONLY") and s < steps - 1:
            if overlap and os.environ.get("DIAG_SIDE_SNAP"):
                with torch.cuda.stream(opt._ov["stream"]):      # ordered after this step's last AdamW, touches no other stream
                    snaps.append((m.flat_params.clone(), m._core._adam_m.clone(), m._core._adam_v.clone(), m._core.shadow.clone()))
            else:
                snaps.append(None)
            continue
        snaps.append((m.flat_params.clone(), opt._ov["core"]._adam_m.clone() if overlap else m._core._adam_m.clone(),
                      m._core._adam_v.clone(), m._core.shadow.clone()))
    torch.cuda.synchronize()
    for s in range(steps):
        if ref is not None and snaps[s] is not None:
            for idx, nm in enumerate(("param", "m", "v", "shadow")):
                d = (snaps[s][idx].float() - ref[s][idx].float()).abs()
                if float(d.max()) > 2e-6:
                    print("MISMATCH step", s, nm, "max", float(d.max()))
                    k = 0
                    for name, off, numel, shape, decay in m._core.tensors:
                        dd = d[off:off + numel]
                        if float(dd.max()) > 2e-6 and k < 4:
                            k += 1
                            print("    ", name, float(dd.max()), float((dd > 2e-6).float().mean()))
                    return snaps, False
    return snaps, True

if os.environ.get("DIAG_STREAM"):
    _S = torch.cuda.Stream()
    torch.cuda.set_stream(_S)
_fo = os.environ.pop("DIAG_FINAL_ONLY", None)
ref, _ = run(False)
if _fo:
    os.environ["DIAG_FINAL_ONLY"] = _fo
bad = 0
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
    _, ok = run(True, ref)
    bad += (not ok)
    if not ok and bad >= 3:
        break
print("bad runs:", bad)
