"""Run the same 3-step (accumulation 2) bf16 trajectory N times; report which parameter tensors differ between runs and by how much.
(diagnostic, GPU)  usage: determinism_probe.py [mode: py|call|graph] [N]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from bert_multimodal_transformer_amd import (AdamW, BertConfig, MAG_BertForSequenceClassification, MultimodalConfig,
                                             get_linear_schedule_with_warmup)
from bert_multimodal_transformer_amd.multimodal_driver import optimizer_grouped_parameters
from oracle import weights
mode = sys.argv[1] if len(sys.argv) > 1 else "py"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 12
graph = {"py": False, "call": None, "graph": True}[mode]
DEV = "cuda:0"

def tb(b):
    t = lambda k: torch.from_numpy(b[k]).to(DEV)
    return t("input_ids"), t("visual"), t("acoustic"), t("input_mask"), t("segment_ids"), t("label_ids")

def run():
    torch.manual_seed(77)
    cfg = BertConfig(num_hidden_layers=2, num_labels=1)
    m = MAG_BertForSequenceClassification(cfg, MultimodalConfig(1.0, 0.5), compute_dtype=torch.bfloat16)
    m.load_state_dict({n: torch.from_numpy(weights.make_param(n, tuple(p.shape), "test")) for n, p in m.named_parameters()})
    m.train()
    opt = AdamW(optimizer_grouped_parameters(m), lr=1e-3)
    sch = get_linear_schedule_with_warmup(opt, 1.0, 10)
    snaps = []
    with m.stream_scope():
        for s in range(6):
            batch = tb(weights.synthetic_bert_batch(4, 32, 47, 74, seed=90 + s))
            upd = (s + 1) % 2 == 0
            m.train_step(*batch, optimizer=opt if upd else None, loss_scale=0.5, graph=graph)
            if upd:
                sch.step()
                snaps.append(("params after update %d" % (s // 2), m.flat_params.clone()))
            else:
                snaps.append(("grads after micro-step %d" % s, m.flat_grads.clone()))
    torch.cuda.synchronize()
    return m, snaps, m.flat_params.clone()

m0, g0, p0 = run()
names = [(n, off, numel) for n, off, numel, shape, dec in m0._core.tensors]
for it in range(1, N):
    m, g, p = run()
    first = None
    for k, ((tag, a), (_, b)) in enumerate(zip(g, g0)):
        if float((a - b).abs().max()) > 2e-6:
            first = k
            break
    print("run %d: max|dparam| %.3e ; first divergence: %s" % (it, float((p - p0).abs().max()), "none" if first is None else g[first][0]))
    if first is not None:
        d = (g[first][1] - g0[first][1]).abs()
        for n, off, numel in names:
            x = float(d[off: off + numel].max())
            if x > 2e-6:
                print("    %-58s max|d| %.3e (max|ref| %.3e) elements %d / %d" % (n, x, float(g0[first][1][off: off + numel].abs().max()),
                                                                                 int((d[off: off + numel] > 2e-6).sum()), numel))
