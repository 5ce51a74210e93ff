"""Numerics of the LayerNorm fold the round-3 review asked to check BEFORE building it: instead of materialising
x = LN(y) in bf16 and feeding it to the next GEMM, keep the un-normalised sum y in bf16, fold gamma into the consumer's weight
(W' = bf16(gamma * W), s = W' 1, c = W beta + b) and finish in the consumer's epilogue:  out = rstd * (y W'^T - mu * s) + c.
The question was the bf16 cancellation in (acc - mu * s).  CPU only: activations and weights come from the fp32 oracle
(BertSelfOutput / BertOutput under /root/reference/bert.py:221-229), the two bf16 paths are emulated with fp32-accumulating matmuls
over bf16-rounded operands, the reference is fp64.  The fold was NOT built (the reasons are structural); this file is the
measurement the decision cites."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import mag_bert_ref as ref          # noqa: E402
from oracle import weights                      # noqa: E402


def _bf(x):
    return x.to(torch.bfloat16).to(torch.float32)


@torch.no_grad()
def _paths(y, gamma, beta, eps, W, b):
    """relative Frobenius error of the standard and of the folded bf16 path against fp64"""
    yd, gd, bd, Wd = y.double(), gamma.double(), beta.double(), W.double()
    mu = yd.mean(-1, keepdim=True)
    rstd = (yd.var(-1, unbiased=False, keepdim=True) + eps).rsqrt()
    want = ((yd - mu) * rstd * gd + bd) @ Wd.t() + b.double()
    # standard: LN in fp32 -> bf16 activations, bf16 weights, fp32 accumulation
    mu32, rstd32 = mu.float(), rstd.float()
    x = _bf((y - mu32) * rstd32 * gamma + beta)
    std = x @ _bf(W).t() + b
    # folded: bf16 y (what the producing GEMM's epilogue stores), statistics from its fp32 accumulators, gamma folded into the weights
    Wg = _bf(gamma * W)
    s = Wg.sum(-1)
    c = W @ beta + b
    fold = rstd32 * (_bf(y) @ Wg.t() - mu32 * s) + c
    n = want.norm()
    return float((std.double() - want).norm() / n), float((fold.double() - want).norm() / n), float((mu.abs() * rstd).mean())


def test_layernorm_fold_bf16_error_against_the_materialised_path():
    torch.manual_seed(0)
    cfg = ref.BertConfigLite(num_hidden_layers=4)
    m = ref.load_deterministic(ref.MAG_BertForSequenceClassification(cfg, ref.MultimodalConfig(1.0, 0.5)), "test").eval()
    with torch.no_grad():                      # LayerNorm parameters away from (1, 0), as after fine-tuning
        for name, p in m.named_parameters():
            if name.endswith("LayerNorm.weight"):
                p.copy_(1.0 + 0.3 * torch.randn_like(p))
            elif name.endswith("LayerNorm.bias"):
                p.copy_(0.3 * torch.randn_like(p))
    caught = {}
    hooks = []
    for li, lyr in enumerate(m.bert.encoder.layer):
        hooks.append(lyr.attention.output.LayerNorm.register_forward_hook(lambda mod, inp, out, k=("ln1", li): caught.__setitem__(k, inp[0].detach())))
        hooks.append(lyr.output.LayerNorm.register_forward_hook(lambda mod, inp, out, k=("ln2", li): caught.__setitem__(k, inp[0].detach())))
    b = weights.synthetic_bert_batch(8, 50, 47, 74, seed=5)
    t = lambda k: torch.from_numpy(b[k])
    with torch.no_grad():
        m(t("input_ids"), t("visual"), t("acoustic"), attention_mask=t("input_mask"), token_type_ids=t("segment_ids"))
    for h in hooks:
        h.remove()
    rows = []
    for li, lyr in enumerate(m.bert.encoder.layer):
        ln1, ln2 = lyr.attention.output.LayerNorm, lyr.output.LayerNorm
        y1 = caught[("ln1", li)].reshape(-1, cfg.hidden_size)
        rows.append(("layer %d LN1 -> FFN-1" % li,) + _paths(y1, ln1.weight, ln1.bias, ln1.eps, lyr.intermediate.dense.weight, lyr.intermediate.dense.bias))
        if li + 1 < len(m.bert.encoder.layer):
            nxt = m.bert.encoder.layer[li + 1].attention.self
            Wqkv = torch.cat([nxt.query.weight, nxt.key.weight, nxt.value.weight], 0)
            bqkv = torch.cat([nxt.query.bias, nxt.key.bias, nxt.value.bias], 0)
            y2 = caught[("ln2", li)].reshape(-1, cfg.hidden_size)
            rows.append(("layer %d LN2 -> QKV" % li,) + _paths(y2, ln2.weight, ln2.bias, ln2.eps, Wqkv, bqkv))
    # stress: what pretrained encoders show -- a row mean far from zero and a few channels 40 sigma out (same weights)
    lyr = m.bert.encoder.layer[1]
    ys = caught[("ln1", 1)].reshape(-1, cfg.hidden_size).clone()
    ys += 3.0 * ys.std()
    ys[:, [10, 300, 511]] += 40.0 * ys.std()
    ln1 = lyr.attention.output.LayerNorm
    rows.append(("stress: mean 3 sigma, three channels +40 sigma",) + _paths(ys, ln1.weight, ln1.bias, ln1.eps, lyr.intermediate.dense.weight, lyr.intermediate.dense.bias))
    for name, e_std, e_fold, mr in rows:
        print("%-48s  materialised bf16 LN: %.3e   folded: %.3e   (|mu| * rstd = %.2f)" % (name, e_std, e_fold, mr))
    # what DESIGN 4.2 states: the fold costs no accuracy (measured: 2.25e-3 against 2.4e-3 for the materialised path on the oracle's
    # activations; 2.8e-3 against 2.5e-3 when |mu| * rstd = 1.2 and three channels sit 40 sigma out) -- the cancellation is not the
    # obstacle, the structure is
    for name, e_std, e_fold, mr in rows[:-1]:
        assert e_fold <= 1.25 * e_std + 1e-4, (name, e_std, e_fold)
    assert rows[-1][2] <= 1.5 * rows[-1][1]
