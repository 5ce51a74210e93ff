import sys, torch
sys.path.insert(0, ".")
from bert_multimodal_transformer_amd import MAG_XLNetModel, MultimodalConfig, XLNetConfig
from oracle import mag_xlnet_ref as X, weights
DEV = "cuda:0"
def tb(b, dev="cpu"):
    t = lambda k: torch.from_numpy(b[k]).to(dev)
    return t("input_ids"), t("visual"), t("acoustic"), t("input_mask"), t("segment_ids"), t("label_ids")
layers, B, L, H = 2, 3, 24, 768
b = weights.synthetic_xlnet_batch(B, L, 47, 74, seed=83)
ids, vis, aco, mask, seg, lab = tb(b, DEV)
i2, v2, a2, m2, s2, l2 = tb(b)
for mode in ("eval", "train"):
    cfg = XLNetConfig(n_layer=layers, dropout=0.0, summary_last_dropout=0.0)
    base = MAG_XLNetModel(cfg, MultimodalConfig(1.0, 0.0), 47, 74)
    ob = X.set_dropout(X.MAG_XLNetModel(X.XLNetConfigLite(n_layer=layers), X.MultimodalConfig(1.0, 0.0), 47, 74), 0.0, 0.0)
    sd = {n: torch.from_numpy(weights.make_param("transformer." + n, tuple(q.shape), "test")) for n, q in base.named_parameters()}
    base.load_state_dict(sd); ob.load_state_dict(sd)
    if mode == "train": base.train(); ob.train()
    else: base.eval(); ob.eval()
    with torch.no_grad():
        out, hs = base(ids, vis, aco, attention_mask=mask, token_type_ids=seg, output_hidden_states=True)
        ro = ob(i2, v2, a2, m2, s2)
    so = base._core.sequence_output(B, L)
    print(mode, "out-ref", float((out.cpu() - ro).abs().max()), "seqout-ref", float((so.cpu() - ro).abs().max()), "hs[-1]-ref", float((hs[-1].cpu() - ro).abs().max()),
          "hs0-out", float((hs[0].cpu()-out.cpu()).abs().max()))
