"""GPU: whole-path parity of MAG_XLNetForSequenceClassification (HIP engine) against
  (a) golden logits / loss produced by the reference's own xlnet.py (tests/golden/g6_xlnet.npz), and
  (b) the CPU oracle (oracle/mag_xlnet_ref.py) run live on the same inputs: gradients, dropout with mask replay,
      optimizer trajectory.
Tolerances as in test_model_gpu.py: fp32 parity mode logits <= 1e-3 (north_star); bf16 stated per test.
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from bert_multimodal_transformer_amd import (AdamW, MAG_XLNetForSequenceClassification, MAG_XLNetModel, MultimodalConfig,
                                             XLNetConfig, get_linear_schedule_with_warmup, rng)
from oracle import mag_xlnet_ref as X
from oracle import optim_ref as O
from oracle import weights

DEV = "cuda:0"


def build(layers=12, cdt=torch.float32, p_mag=0.5, p=0.1, mode="test", V=47, mem_len=None):
    cfg = XLNetConfig(n_layer=layers, num_labels=1, dropout=p, summary_last_dropout=p, mem_len=mem_len)
    m = MAG_XLNetForSequenceClassification(cfg, MultimodalConfig(1.0, p_mag), visual_dim=V, acoustic_dim=74, compute_dtype=cdt)
    sd = {n: torch.from_numpy(weights.make_param(n, tuple(q.shape), mode)) for n, q in m.named_parameters()}
    m.load_state_dict(sd)
    return m


def oracle(layers=12, p_mag=0.5, mode="test", V=47):
    o = X.MAG_XLNetForSequenceClassification(X.XLNetConfigLite(n_layer=layers), X.MultimodalConfig(1.0, p_mag), V, 74)
    return X.load_deterministic(o, mode)


def tb(b, dev="cpu"):
    t = lambda k: torch.from_numpy(b[k]).to(dev)
    return t("input_ids"), t("visual"), t("acoustic"), t("input_mask"), t("segment_ids"), t("label_ids")


LOOSE_BF16 = ("MAG.W_hv", "MAG.W_ha", "MAG.W_v", "MAG.W_a")      # relu / clamp gated: single elements flip on a bf16 pre-activation


def _grad_report(m, o, tol, frobenius=False, loose=(), tol_loose=None, show=0):
    og = {n: p.grad for n, p in o.named_parameters() if p.grad is not None}
    gmax = max(float(g.abs().max()) for g in og.values())
    gnorm = max(float(g.norm()) for g in og.values())
    rows = []
    for n, p in m.named_parameters():
        if n not in og:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, n
            continue
        g, r = p.grad.detach().cpu(), og[n]
        if frobenius:
            rel = float((g - r).norm()) / max(float(r.norm()), 1e-3 * gnorm)
        else:
            rel = float((g - r).abs().max()) / max(float(r.abs().max()), 1e-3 * gmax)
        rows.append((rel if rel == rel else float("inf"), n))
    rows.sort(reverse=True)
    print("worst relative gradient error %.3e at %s" % rows[0])
    for rel, n in rows[:show]:
        print("    %.3e  %s" % (rel, n))
    for rel, n in rows:
        t = tol_loose if (tol_loose is not None and any(k in n for k in loose)) else tol
        assert rel <= t, (rel, n, t)


def test_state_dict_and_frozen_mask_emb():
    m, o = build(layers=2), oracle(layers=2)
    assert set(m.state_dict().keys()) == set(o.state_dict().keys())
    for k, v in o.state_dict().items():
        assert torch.equal(m.state_dict()[k].cpu(), v), k
    assert m.transformer.mask_emb.grad is None
    # reference grouping rule (multimodal_driver.py:328-343): "layer_norm.weight" IS decayed (only "LayerNorm" is matched)
    from bert_multimodal_transformer_amd.multimodal_driver import optimizer_grouped_parameters
    groups = optimizer_grouped_parameters(m)
    names = {id(p): n for n, p in m.named_parameters()}
    assert "transformer.layer.0.rel_attn.layer_norm.weight" in {names[id(p)] for p in groups[0]["params"]}
    assert "transformer.layer.0.rel_attn.r_r_bias" in {names[id(p)] for p in groups[1]["params"]}


@pytest.mark.parametrize("B,L,seed", [(4, 50, 31), (48, 50, 32), (3, 128, 34), (2, 100, 35)])
def test_eval_logits_match_reference_golden_fp32(golden, B, L, seed):
    m = build().eval()
    ids, vis, aco, mask, seg, _ = tb(weights.synthetic_xlnet_batch(B, L, 47, 74, seed=seed), DEV)
    with torch.no_grad():
        logits = m(ids, vis, aco, token_type_ids=seg, attention_mask=mask, labels=None)[0]
    ref = golden["g6_xlnet"]["logits/B%d_L%d_seed%d" % (B, L, seed)]
    err = float(np.abs(logits.cpu().numpy() - ref).max())
    print("xlnet fp32 eval logits max|err| = %.3e (|logit| max %.3f)" % (err, float(np.abs(ref).max())))
    assert err <= 1e-3


def test_eval_logits_bf16(golden):
    m = build(cdt=torch.bfloat16).eval()
    ids, vis, aco, mask, seg, _ = tb(weights.synthetic_xlnet_batch(48, 50, 47, 74, seed=32), DEV)
    with torch.no_grad():
        logits = m(ids, vis, aco, token_type_ids=seg, attention_mask=mask)[0]
    ref = golden["g6_xlnet"]["logits/B48_L50_seed32"]
    err = float(np.abs(logits.cpu().numpy() - ref).max())
    print("xlnet bf16 eval logits max|err| = %.3e (|logit| max %.3f)" % (err, float(np.abs(ref).max())))
    assert err <= 5e-2


@pytest.mark.parametrize("B,L", [(3, 17), (2, 64), (5, 33), (2, 65), (3, 100), (2, 128), (1, 97)])
def test_ragged_shapes_eval_fp32(B, L):
    """sequence lengths that are not multiples of 16, both sides of the L = 64 kernel boundary, the L = 128 maximum, pad-free rows"""
    m = build(layers=2).eval()
    o = oracle(layers=2).eval()
    b = weights.synthetic_xlnet_batch(B, L, 47, 74, seed=70 + L)
    b["input_mask"][0, :] = 1                                    # one row without padding
    ids, vis, aco, mask, seg, _ = tb(b, DEV)
    with torch.no_grad():
        l1 = m(ids, vis, aco, token_type_ids=seg, attention_mask=mask)[0].cpu()
        l0 = o(*tb(b)[:5])[0]
    err = float((l1 - l0).abs().max())
    print("B=%d L=%d max|err| %.3e" % (B, L, err))
    assert err <= 1e-3


def test_base_model_sequence_output_fp32():
    cfg = XLNetConfig(n_layer=2)
    m = MAG_XLNetModel(cfg, MultimodalConfig(1.0, 0.5), 47, 74).eval()
    o = X.MAG_XLNetModel(X.XLNetConfigLite(n_layer=2), X.MultimodalConfig(1.0, 0.5), 47, 74).eval()
    sd = {n: torch.from_numpy(weights.make_param("transformer." + n, tuple(q.shape), "test")) for n, q in m.named_parameters()}
    m.load_state_dict(sd); o.load_state_dict(sd)
    b = weights.synthetic_xlnet_batch(4, 50, 47, 74, seed=81)
    ids, vis, aco, mask, seg, _ = tb(b, DEV)
    with torch.no_grad():
        y1 = m(ids, vis, aco, attention_mask=mask, token_type_ids=seg)[0].cpu()
        y0 = o(*tb(b)[:5])
    err = float((y1 - y0).abs().max())
    print("sequence output max|err| %.3e (max %.3f)" % (err, float(y0.abs().max())))
    assert err <= 1e-3


def test_inputs_embeds_and_trainable_base_model_fp32():
    """f-4 for MAG-XLNet: inputs_embeds (xlnet.py:201-213, 301-305) replaces the table gather and receives its own gradient;
    MAG_XLNetModel's output (xlnet.py:396-405) carries an autograd edge into the engine -- a user head on it trains the whole
    stack (gradients vs the oracle); in train mode that output is the last hidden state under the final dropout's mask."""
    layers, B, L, H = 2, 3, 24, 768
    # (a) classification model, inputs_embeds with its gradient
    m = build(layers, p_mag=0.0, p=0.0).train()
    o = X.set_dropout(oracle(layers, p_mag=0.0), 0.0, 0.0).train()
    b = weights.synthetic_xlnet_batch(B, L, 47, 74, seed=83)
    ids, vis, aco, mask, seg, lab = tb(b, DEV)
    i2, v2, a2, m2, s2, l2 = tb(b)
    emb_cpu = (o.transformer.word_embedding(i2).detach() + 0.05 * torch.from_numpy(weights.make_param("probe.emb", (B, L, H), "test"))).requires_grad_(True)
    emb = emb_cpu.detach().to(DEV).requires_grad_(True)
    logits = m(None, vis, aco, attention_mask=mask, token_type_ids=seg, inputs_embeds=emb)[0]
    torch.nn.MSELoss()(logits.view(-1), lab.view(-1)).backward()
    lo = o(None, v2, a2, m2, s2, inputs_embeds=emb_cpu)[0]
    F.mse_loss(lo.view(-1), l2.view(-1)).backward()
    torch.cuda.synchronize()
    err = float((logits.detach().cpu() - lo.detach()).abs().max())
    gerr = float((emb.grad.cpu() - emb_cpu.grad).norm() / emb_cpu.grad.norm())
    print("xlnet inputs_embeds: logits %.2e, d(inputs_embeds) rel. Frobenius %.2e" % (err, gerr))
    assert err <= 1e-3 and gerr <= 2e-3
    word = dict(m.named_parameters())["transformer.word_embedding.weight"]
    assert float(word.grad.abs().max()) == 0.0 and o.transformer.word_embedding.weight.grad is None
    o.transformer.word_embedding.weight.grad = torch.zeros_like(o.transformer.word_embedding.weight)
    _grad_report(m, o, 5e-3)
    with pytest.raises(ValueError):
        m(ids, vis, aco, inputs_embeds=emb)
    with pytest.raises(ValueError):
        m(None, vis, aco)
    # nothing sticks: the ids path afterwards equals the oracle's, and the single-call step runs
    m.eval(); o.eval()
    with torch.no_grad():
        assert float((m(ids, vis, aco, attention_mask=mask, token_type_ids=seg)[0].cpu() - o(i2, v2, a2, m2, s2)[0]).abs().max()) <= 1e-3
    m.train()
    m.zero_grad()
    m.train_step(ids, vis, aco, mask, seg, lab, optimizer=None)
    torch.cuda.synchronize()
    assert float(word.grad.abs().max()) > 0.0
    # (b) base model: a head on its output back-propagates into the engine
    cfg = XLNetConfig(n_layer=layers, dropout=0.0, summary_last_dropout=0.0)
    base = MAG_XLNetModel(cfg, MultimodalConfig(1.0, 0.0), 47, 74).train()
    ob = X.set_dropout(X.MAG_XLNetModel(X.XLNetConfigLite(n_layer=layers), X.MultimodalConfig(1.0, 0.0), 47, 74), 0.0, 0.0).train()
    sd = {n: torch.from_numpy(weights.make_param("transformer." + n, tuple(q.shape), "test")) for n, q in base.named_parameters()}
    base.load_state_dict(sd); ob.load_state_dict(sd)
    out = base(ids, vis, aco, attention_mask=mask, token_type_ids=seg)[0]
    assert out.requires_grad
    w = torch.from_numpy(weights.make_param("probe.seq", (H,), "test"))
    ((out * w.to(DEV)).sum(-1) ** 2).mean().backward()
    ro = ob(i2, v2, a2, m2, s2)
    ((ro * w).sum(-1) ** 2).mean().backward()
    torch.cuda.synchronize()
    assert float((out.detach().cpu() - ro.detach()).abs().max()) <= 1e-3
    og = {n: q.grad for n, q in ob.named_parameters() if q.grad is not None}
    gmax = max(float(g.abs().max()) for g in og.values())
    rows = sorted(((float((q.grad.cpu() - og[n]).abs().max()) / max(float(og[n].abs().max()), 1e-3 * gmax), n)
                   for n, q in base.named_parameters() if n in og), reverse=True)
    print("base-model gradients through the autograd edge: worst relative errors", ["%.2e %s" % r for r in rows[:4]])
    assert rows[0][0] <= 5e-3
    # train mode with dropout: the returned tensor is the last hidden state under the final dropout's counter-hash mask
    cfg = XLNetConfig(n_layer=layers, dropout=0.1, summary_last_dropout=0.1)
    base = MAG_XLNetModel(cfg, MultimodalConfig(1.0, 0.5), 47, 74).train()
    base.load_state_dict(sd)
    with torch.no_grad():
        out, hs = base(ids, vis, aco, attention_mask=mask, token_type_ids=seg, output_hidden_states=True)
    mult = torch.from_numpy(rng.keep_mult(B * L * H, rng.make_key(base._core.seed, base._core.step, rng.XS_FINAL, 0.1))).view(B, L, H)
    assert float((out.cpu() - hs[-1].cpu() * mult).abs().max()) <= 1e-6
    assert 0.05 < float((mult == 0).float().mean()) < 0.15


def test_output_attentions_and_hidden_states_fp32():
    """f-4 for MAG-XLNet (xlnet.py:363-427): output_attentions = the probabilities the forward keeps for its backward, one
    [B, n_head, L, L] tensor per layer, on the base model and passed through by the classification model (eval mode here; the
    train-mode tensors -- after the attention dropout -- are compared in test_train_mode_dropout_mask_replay)."""
    layers, B, L, nh = 2, 3, 40, 12
    m = MAG_XLNetForSequenceClassification(XLNetConfig(n_layer=layers), MultimodalConfig(1.0, 0.5), 47, 74).eval()
    m.load_state_dict({n: torch.from_numpy(weights.make_param(n, tuple(q.shape), "test")) for n, q in m.named_parameters()})
    o = X.load_deterministic(X.MAG_XLNetForSequenceClassification(X.XLNetConfigLite(n_layer=layers), X.MultimodalConfig(1.0, 0.5), 47, 74)).eval()
    b = weights.synthetic_xlnet_batch(B, L, 47, 74, seed=83)
    ids, vis, aco, mask, seg, _ = tb(b, DEV)
    with torch.no_grad():
        out = m(ids, vis, aco, token_type_ids=seg, attention_mask=mask, output_attentions=True, output_hidden_states=True)
        o(*tb(b)[:5])
        base = m.transformer(ids, vis, aco, token_type_ids=seg, attention_mask=mask, output_attentions=True)
    assert len(out) == 3 and len(out[1]) == layers + 1 and len(out[2]) == layers
    for l in range(layers):
        a, r = out[2][l].cpu(), o.transformer.layer[l].rel_attn.last_probs
        assert tuple(a.shape) == (B, nh, L, L)
        err = float((a - r).abs().max())
        print("layer %d attention probabilities max|err| %.3e" % (l, err))
        assert err <= 1e-5
        assert float((a.sum(-1) - 1.0).abs().max()) <= 1e-4
    assert len(base) == 2 and float((base[1][0] - out[2][0]).abs().max()) == 0.0


@pytest.mark.parametrize("cdt", [torch.float32, torch.bfloat16])
def test_head_mask_vs_oracle(cdt):
    """f-4 for MAG-XLNet (xlnet.py:336-353, 383): head_mask [n_layer][n_head] (and the 1-D form) scales attn_prob after the
    dropout.  Train mode, every dropout p = 0: logits, the returned attention probabilities and every parameter gradient against
    the oracle; afterwards a plain forward is unaffected and the single-call step runs."""
    layers, B, L, nh = 2, 3, 24, 12
    fp32 = cdt == torch.float32
    m = build(layers, cdt, p_mag=0.0, p=0.0).train()
    o = X.set_dropout(oracle(layers, p_mag=0.0), 0.0, 0.0).train()
    b = weights.synthetic_xlnet_batch(B, L, 47, 74, seed=47)
    ids, vis, aco, mask, seg, lab = tb(b, DEV)
    hm = torch.ones(layers, nh)
    hm[0, 2] = 0.0; hm[0, 9] = 0.5; hm[1, 0] = 0.0; hm[1, 5] = 2.0
    out = m(ids, vis, aco, token_type_ids=seg, attention_mask=mask, head_mask=hm.to(DEV), output_attentions=True)
    logits, att = out[0], out[1]
    torch.nn.MSELoss()(logits.view(-1), lab.view(-1)).backward()
    i2, v2, a2, m2, s2, l2 = tb(b)
    lo = o(i2, v2, a2, m2, s2, head_mask=hm)[0]
    torch.nn.functional.mse_loss(lo.view(-1), l2.view(-1)).backward()
    torch.cuda.synchronize()
    err = float((logits.detach().cpu() - lo.detach()).abs().max())
    perr = max(float((att[l].cpu() - lyr.rel_attn.last_probs.detach()).abs().max()) for l, lyr in enumerate(o.transformer.layer))
    print("xlnet head_mask (%s): logits %.2e, probabilities %.2e" % (cdt, err, perr))
    assert err <= (1e-3 if fp32 else 5e-2) and perr <= (1e-5 if fp32 else 2e-2)
    assert float(att[0][:, 2].abs().max()) == 0.0 and float(att[1][:, 0].abs().max()) == 0.0
    _grad_report(m, o, 5e-3 if fp32 else 1e-1, frobenius=not fp32)
    m.eval(); o.eval()
    with torch.no_grad():
        plain = m(ids, vis, aco, token_type_ids=seg, attention_mask=mask)[0].cpu()
        one = m(ids, vis, aco, token_type_ids=seg, attention_mask=mask, head_mask=hm[0].to(DEV))[0].cpu()       # 1-D: every layer
        assert float((plain - o(i2, v2, a2, m2, s2)[0]).abs().max()) <= (1e-3 if fp32 else 5e-2)
        assert float((one - o(i2, v2, a2, m2, s2, head_mask=hm[0])[0]).abs().max()) <= (1e-3 if fp32 else 5e-2)
    m.train(); m.zero_grad()
    m.train_step(ids, vis, aco, mask, seg, lab, optimizer=None)
    torch.cuda.synchronize()


def test_gradients_match_oracle_fp32(golden):
    """train mode, every dropout p = 0: loss and all parameter gradients vs the oracle (and the golden loss)."""
    m = build(p_mag=0.0, p=0.0).train()
    o = X.set_dropout(oracle(p_mag=0.0), 0.0, 0.0).train()
    b = weights.synthetic_xlnet_batch(4, 50, 47, 74, seed=33)
    ids, vis, aco, mask, seg, lab = tb(b, DEV)
    logits = m(ids, vis, aco, token_type_ids=seg, attention_mask=mask, labels=None)[0]
    loss = torch.nn.MSELoss()(logits.view(-1), lab.view(-1))
    loss.backward()
    i2, v2, a2, m2, s2, l2 = tb(b)
    lo = torch.nn.functional.mse_loss(o(i2, v2, a2, m2, s2)[0].view(-1), l2.view(-1))
    lo.backward()
    torch.cuda.synchronize()
    assert abs(float(loss.detach()) - float(golden["g6_xlnet"]["train/loss_B4_L50_seed33"])) < 1e-4
    assert abs(float(loss.detach()) - float(lo.detach())) < 1e-4
    _grad_report(m, o, 2e-3)
    for n, p in m.named_parameters():                       # and against the reference's own gradient samples
        if p.grad is not None:
            np.testing.assert_allclose(float(p.grad.norm()), float(golden["g6_xlnet"]["train/gnorm/" + n]), rtol=5e-3, atol=1e-5)


def test_fused_training_step_equals_autograd_path():
    m = build(layers=2, p_mag=0.0, p=0.0).train()
    ids, vis, aco, mask, seg, lab = tb(weights.synthetic_xlnet_batch(6, 50, 47, 74, seed=34), DEV)
    logits = m(ids, vis, aco, token_type_ids=seg, attention_mask=mask, labels=None)[0]
    loss = torch.nn.MSELoss()(logits.view(-1), lab.view(-1))
    loss.backward()
    g_ref = m.flat_grads.clone()
    m.zero_grad()
    l2 = m.training_step(ids, vis, aco, mask, seg, lab)
    assert abs(float(l2) - float(loss.detach())) < 1e-5
    assert float((m.flat_grads - g_ref).abs().max()) <= 1e-5 * float(g_ref.abs().max()) + 1e-9


class _SeqReplay(torch.nn.Module):
    """dropout module that multiplies its k-th call by the k-th host-regenerated device mask"""

    def __init__(self, mults):
        super().__init__()
        self.mults, self.k = mults, 0

    def forward(self, x):
        mlt = self.mults[self.k % len(self.mults)]
        self.k += 1
        assert mlt.shape == x.shape, (mlt.shape, x.shape)
        return x * mlt


def oracle_replay_grads(layers, B, L, seed, step, b, double=False, probe=None, flip=None):
    """flip = (gate, index) or a list of such pairs: the sign of those pre-activations of MAG's relu gate `gate` ("W_hv" | "W_ha") is
    inverted (|x| ~ 1e-7: the forward does not change, the gate's derivative does)"""
    o = oracle(layers).train()
    if double:
        o = o.double()
    if flip is not None:
        flips = [flip] if isinstance(flip[0], str) else list(flip)          # one (gate, index) or a list of them
        for gate in ("W_hv", "W_ha"):
            idxs = [tuple(f[1]) for f in flips if f[0] == gate]
            if not idxs:
                continue

            def _flip(mod, args, out, idxs=idxs):
                out = out.clone()
                for idx in idxs:          # value -> -value with d(out)/d(pre) still +1: a plain negation would also negate the gate's
                    out[idx] = out[idx] - 2.0 * out[idx].detach()          # derivative when it OPENS a closed gate (round 6, draw 63)
                return out
            getattr(o.transformer.MAG, gate).register_forward_hook(_flip)
    if probe is not None:          # the pre-activations of MAG's relu gates (modeling.py:27-28) ...
        for lin in (o.transformer.MAG.W_hv, o.transformer.MAG.W_ha):
            lin.register_forward_hook(lambda mod, args, out: probe.append(out.detach()))

        def _clamp_margin(mod, args):          # ... and the argument of the other kink, alpha = min(|e| / (|h_m| + eps) * beta, 1) (modeling.py:32-43)
            e, v, a = (x.detach() for x in args)
            lin = lambda layer, x: F.linear(x, layer.weight, layer.bias)          # (not layer(x): that would fire the probes' hooks)
            wv = torch.relu(lin(mod.W_hv, torch.cat((v, e), dim=-1)))
            wa = torch.relu(lin(mod.W_ha, torch.cat((a, e), dim=-1)))
            hm = (wv * lin(mod.W_v, v) + wa * lin(mod.W_a, a)).norm(2, dim=-1)
            hm = torch.where(hm == 0, torch.ones_like(hm), hm)
            probe.append((e.norm(2, dim=-1) / (hm + 1e-6) * mod.beta_shift - 1.0).detach())
        o.transformer.MAG.register_forward_pre_hook(_clamp_margin)
    nh, H, DI = 12, 768, 3072
    cast = (lambda t: t.double()) if double else (lambda t: t)
    mult = lambda site, p, n: cast(torch.from_numpy(rng.keep_mult(n, rng.make_key(seed, step, site, p))))
    blx = lambda site, p, Xd: mult(site, p, B * L * Xd).view(B, L, Xd).permute(1, 0, 2)
    S = _SeqReplay
    o.transformer.dropout = S([blx(rng.XS_EMB, 0.1, H), mult(rng.XS_POS, 0.1, 2 * L * B * H).view(2 * L, B, H), blx(rng.XS_FINAL, 0.1, H)])
    o.transformer.MAG.dropout = S([blx(rng.XS_MAG, 0.5, H)])
    o.sequence_summary.last_dropout = S([mult(rng.XS_HEAD, 0.1, B * H).view(B, H)])
    for l, lyr in enumerate(o.transformer.layer):
        s0 = rng.XS_LAYER0 + 8 * l
        lyr.rel_attn.dropout = S([mult(s0 + 0, 0.1, B * nh * L * L).view(B, nh, L, L), blx(s0 + 1, 0.1, H)])
        lyr.ff.dropout = S([blx(s0 + 2, 0.1, DI), blx(s0 + 3, 0.1, H)])
    i2, v2, a2, m2, s2, l2 = tb(b)
    if double:
        v2, a2, l2 = v2.double(), a2.double(), l2.double()
    lo = o(i2, v2, a2, m2, s2)[0]
    torch.nn.functional.mse_loss(lo.view(-1), l2.view(-1)).backward()
    return {n: p.grad for n, p in o.named_parameters() if p.grad is not None}, lo.detach()


@pytest.mark.parametrize("draw", [81, 64])
def test_mag_gate_kink_explains_gradient_outliers_fp32(draw):
    """Round 4's once-seen 7.9e-2 at MAG.W_ha (identical logits) was not a race: MAG's relu gates (modeling.py:27-28) have a kink at
    zero, and about one dropout draw in forty puts a pre-activation within fp32 rounding of it -- the GPU and the CPU then take
    different sides of a discontinuous derivative and ONE token's contribution to dW_hv / dW_ha flips.  This is draw (seed 12345, step
    81) of scripts/exp/flake_hunt.py --vary (profiles/r05_flake_hunt.txt): against a float64 oracle the GPU gradient is off by 6e-2 of
    the tensor's maximum as it stands and by 2e-6 once the sign of ONE gate pre-activation (|x| = 4.6e-7) is inverted in the oracle.
    The assertion holds whichever side a future kernel lands on: the GPU gradient must equal the exact gradient for one of the two
    states of a gate whose pre-activation is within 1e-5 of zero.
    Draw 64 (the hunt's "draw 63", 8.6e-3 at MAG.W_ha) was the one round 5 could not explain: there the float64 pre-activation is
    NEGATIVE (W_ha[17,1,143] = -3.5e-7, gate closed) and the GPU has it open, and the hunt's flip hook negated the value with a plain
    `-x`, which also negates the derivative of the gate it opens (profiles/r06_flake_draw63.txt: GPU row - oracle row = -1.000001 x what
    that hook added).  With the derivative kept at +1 the same single gate reproduces the GPU gradient to 1.8e-6."""
    layers, B, L = 2, 3, 24
    torch.manual_seed(12345)
    m = build(layers, torch.float32).train()
    m._core.step = draw - 1
    b = weights.synthetic_xlnet_batch(B, L, 47, 74, seed=41)
    ids, vis, aco, mask, seg, lab = tb(b, DEV)
    out = m(ids, vis, aco, token_type_ids=seg, attention_mask=mask, labels=None)
    torch.nn.MSELoss()(out[0].view(-1), lab.view(-1)).backward()
    torch.cuda.synchronize()
    seed, step = m._core.seed, m._core.step
    assert (seed, step) == (12345, draw)
    probe = []
    o64, lo = oracle_replay_grads(layers, B, L, seed, step, b, double=True, probe=probe)
    probe = probe[1:]                                        # [clamp margin, W_hv, W_ha] -> the two gates
    assert float((out[0].detach().cpu().double() - lo).abs().max()) <= 1e-5          # the forward does not see the kink
    gmax = max(float(g.abs().max()) for g in o64.values())
    worst = lambda ref: max(float((p.grad.detach().cpu().double() - ref[n]).abs().max()) / max(float(ref[n].abs().max()), 1e-3 * gmax)
                            for n, p in m.named_parameters() if n in ref)
    as_is = worst(o64)
    best = (as_is, None)
    for k in range(2):
        flat = probe[k].abs().flatten()
        for j in torch.topk(flat, 3, largest=False).indices.tolist():
            if float(flat[j]) > 1e-5:
                continue
            idx = [int(i) for i in torch.unravel_index(torch.tensor(j), probe[k].shape)]
            of, _ = oracle_replay_grads(layers, B, L, seed, step, b, double=True, flip=(("W_hv", "W_ha")[k], idx))
            w = worst(of)
            if w < best[0]:
                best = (w, (("W_hv", "W_ha")[k], idx, float(probe[k][tuple(idx)])))
    print("GPU vs float64 oracle, every tensor: %.3e as is; %.3e with the gate state %s" % (as_is, best[0], best[1]))
    assert best[0] <= 5e-5


@pytest.mark.parametrize("cdt,tol_logit,tol_grad,L,layers,B", [
    (torch.float32, 1e-3, 5e-3, 24, 2, 3),
    (torch.bfloat16, 5e-2, 1e-1, 24, 2, 3),      # bf16 gradients: relative Frobenius error, dominated by relu / clamp flips in MAG (measured 7.7e-2 on W_hv)
    (torch.float32, 1e-3, 5e-3, 100, 2, 3),      # ... and above the L = 64 kernel boundary
    (torch.bfloat16, 5e-2, 3e-2, 50, 12, 48),    # the shape bench.py's `secondary` line times: 12 layers, B=48, L=50 (one strip group, 50 -> 64 padding),
    (torch.float32, 1e-3, 5e-3, 50, 12, 48)])    # bf16 (xl_attn_bwd_kv2_kernel) and fp32, through the FUSED training step
def test_train_mode_dropout_mask_replay(cdt, tol_logit, tol_grad, L, layers, B):
    """Dropout ON at every site (0.1 hidden / attention / pos_emb / summary, MAG 0.5): device masks regenerated on the
    host and replayed inside the oracle (which works in the reference's [L, B, .] layout) -> exact train-mode parity.  The
    per-(position, sample) mask of pos_emb (xlnet.py:332-333 drops the [2L, B, H] expansion) is part of the replay.  The 12-layer
    legs run the fused training step (what train_epoch / bench.py call); the 2-layer legs the autograd route with attentions."""
    nh, H, DI = 12, 768, 3072
    full = layers == 12
    # The engine's dropout seed is torch.initial_seed() at model creation: without this line every pytest run draws other masks, and
    # about one draw in a hundred puts a pre-activation of MAG's relu gates (modeling.py:27-28) within fp32 rounding of zero -- the
    # CPU and the GPU then take different sides of a discontinuous derivative: identical logits, one token's contribution to
    # dW_hv / dW_ha flipped (3e-2 .. 8e-2 of the tensor's largest gradient).  That was round 4's "7.9e-2 at MAG.W_ha once in eight
    # runs" (scripts/exp/flake_hunt.py --vary reproduces it at will and compares both sides with a float64 oracle:
    # profiles/r05_flake_hunt.txt).  A pinned seed makes the case the same every run.
    torch.manual_seed(99)
    m = build(layers, cdt).train()
    o = oracle(layers).train()
    core = m._core
    b = weights.synthetic_xlnet_batch(B, L, 47, 74, seed=41)
    ids, vis, aco, mask, seg, lab = tb(b, DEV)
    if full:
        loss = m.training_step(ids, vis, aco, mask, seg, lab)
        logits = att = None
    else:
        out = m(ids, vis, aco, token_type_ids=seg, attention_mask=mask, labels=None, output_attentions=True)
        logits, att = out[0], out[1]
        torch.nn.MSELoss()(logits.view(-1), lab.view(-1)).backward()
    seed, step = core.seed, core.step
    mult = lambda site, p, n: torch.from_numpy(rng.keep_mult(n, rng.make_key(seed, step, site, p)))
    blx = lambda site, p, Xd: mult(site, p, B * L * Xd).view(B, L, Xd).permute(1, 0, 2)        # engine [B,L,X] -> oracle [L,B,X]
    o.transformer.dropout = _SeqReplay([blx(rng.XS_EMB, 0.1, H), mult(rng.XS_POS, 0.1, 2 * L * B * H).view(2 * L, B, H),
                                        blx(rng.XS_FINAL, 0.1, H)])
    o.transformer.MAG.dropout = _SeqReplay([blx(rng.XS_MAG, 0.5, H)])
    o.sequence_summary.last_dropout = _SeqReplay([mult(rng.XS_HEAD, 0.1, B * H).view(B, H)])
    for l, lyr in enumerate(o.transformer.layer):
        s0 = rng.XS_LAYER0 + 8 * l
        lyr.rel_attn.dropout = _SeqReplay([mult(s0 + 0, 0.1, B * nh * L * L).view(B, nh, L, L), blx(s0 + 1, 0.1, H)])
        lyr.ff.dropout = _SeqReplay([blx(s0 + 2, 0.1, DI), blx(s0 + 3, 0.1, H)])
    i2, v2, a2, m2, s2, l2 = tb(b)
    lo = o(i2, v2, a2, m2, s2)[0]
    loss_o = torch.nn.functional.mse_loss(lo.view(-1), l2.view(-1))
    loss_o.backward()
    torch.cuda.synchronize()
    if full:
        print("xlnet train-mode fused step %s: loss %.6f vs oracle %.6f" % (cdt, float(loss), float(loss_o)))
        assert abs(float(loss) - float(loss_o)) <= (2e-5 if cdt == torch.float32 else 5e-3) * max(1.0, abs(float(loss_o)))
    else:
        err = float((logits.detach().cpu() - lo.detach()).abs().max())
        print("xlnet train-mode logits max|err|:", err)
        assert err <= tol_logit
        # output_attentions in train mode: the probabilities AFTER the (replayed) attention dropout
        perr = max(float((att[l].cpu() - lyr.rel_attn.last_probs.detach()).abs().max()) for l, lyr in enumerate(o.transformer.layer))
        print("xlnet train-mode attention probabilities max|err|:", perr)
        assert perr <= (1e-5 if cdt == torch.float32 else 2e-2)
    _grad_report(m, o, tol_grad, frobenius=(cdt == torch.bfloat16), loose=LOOSE_BF16 if full else (), tol_loose=1e-1, show=6 if full else 0)


@pytest.mark.parametrize("B,L", [(48, 50)])
def test_training_step_bf16_full_model_vs_oracle(B, L):
    """The MAG-XLNet configuration bench.py times (BASELINE configs[3]: 12 layers, bf16 perf mode, B=48, L=50, MOSI), one fused
    training step with every dropout p = 0: loss within 2e-3, every gradient tensor within 3e-2 relative Frobenius error of the
    fp32 CPU oracle (MAG's gated tensors 1e-1), the 13 per-layer hidden states within 2e-2.  (xl_attn_bwd_kv2_kernel is the bf16
    L <= 64 key / position-side backward: this is its 12-layer check at the padded 50 -> 64 strip.)"""
    m = build(12, torch.bfloat16, p_mag=0.0, p=0.0).train()
    o = X.set_dropout(oracle(12, p_mag=0.0), 0.0, 0.0).train()
    b = weights.synthetic_xlnet_batch(B, L, 47, 74, seed=71)
    ids, vis, aco, mask, seg, lab = tb(b, DEV)
    loss = m.training_step(ids, vis, aco, mask, seg, lab)
    hs = m._core.hidden_states(B, L)
    hooks, ref_h = [], []
    hooks.append(o.transformer.layer[0].register_forward_pre_hook(lambda mod, args: ref_h.append(args[0].detach())))
    for lyr in o.transformer.layer:
        hooks.append(lyr.register_forward_hook(lambda mod, args, out: ref_h.append(out.detach())))
    i2, v2, a2, m2, s2, l2 = tb(b)
    lo = F.mse_loss(o(i2, v2, a2, m2, s2)[0].view(-1), l2.view(-1))
    lo.backward()
    for h in hooks:
        h.remove()
    torch.cuda.synchronize()
    print("xlnet bf16 B=%d L=%d: loss %.5f vs oracle %.5f" % (B, L, float(loss), float(lo)))
    assert abs(float(loss) - float(lo)) <= 2e-3 * max(1.0, abs(float(lo)))
    assert len(hs) == len(ref_h) == 13
    worst = 0.0
    for h, r in zip(hs, ref_h):                                # engine [B, L, H] vs oracle [L, B, H]
        r = r.permute(1, 0, 2)
        worst = max(worst, float((h.float().cpu() - r).norm() / r.norm()))
    print("hidden states: worst relative Frobenius error over the 13 entries %.3e" % worst)
    assert worst <= 2e-2
    _grad_report(m, o, 3e-2, frobenius=True, loose=LOOSE_BF16, tol_loose=1e-1, show=6)


def test_three_optimizer_steps_track_the_oracle_fp32():
    layers = 2
    m = build(layers=layers, p_mag=0.0, p=0.0).train()
    o = X.set_dropout(oracle(layers=layers, p_mag=0.0), 0.0, 0.0).train()
    from bert_multimodal_transformer_amd.multimodal_driver import optimizer_grouped_parameters
    opt = AdamW(optimizer_grouped_parameters(m), lr=1e-3)
    sch = get_linear_schedule_with_warmup(opt, num_warmup_steps=1.0, num_training_steps=10)
    oo = O.AdamW(O.grouped_parameters(o), lr=1e-3)
    so = O.get_linear_schedule_with_warmup(oo, num_warmup_steps=1.0, num_training_steps=10)
    mask0 = m.transformer.mask_emb.detach().clone()
    for s in range(3):
        b = weights.synthetic_xlnet_batch(4, 50, 47, 74, seed=50 + s)
        ids, vis, aco, mask, seg, lab = tb(b, DEV)
        m.training_step(ids, vis, aco, mask, seg, lab)
        opt.step(); sch.step(); opt.zero_grad()
        i2, v2, a2, m2, s2, l2 = tb(b)
        oo.zero_grad()
        torch.nn.functional.mse_loss(o(i2, v2, a2, m2, s2)[0].view(-1), l2.view(-1)).backward()
        oo.step(); so.step()
    torch.cuda.synchronize()
    assert float(m.flat_grads.abs().max()) == 0.0
    assert torch.equal(m.transformer.mask_emb.detach(), mask0)         # no gradient -> HF AdamW never touches it (no decay either)
    om = dict(o.named_parameters())
    worst = max(float((p.detach().cpu() - om[n].detach()).abs().max()) for n, p in m.named_parameters())
    print("max |param - oracle param| after 3 steps:", worst)
    assert worst <= 2e-4
    m.eval(); o.eval()
    b = weights.synthetic_xlnet_batch(4, 50, 47, 74, seed=60)
    ids, vis, aco, mask, seg, _ = tb(b, DEV)
    with torch.no_grad():
        l1 = m(ids, vis, aco, token_type_ids=seg, attention_mask=mask)[0].cpu()
        l0 = o(*tb(b)[:5])[0]
    assert float((l1 - l0).abs().max()) <= 5e-3


# (fp32 at L = 128: worst tensor is MAG.W_ha at 6.7e-3 of its max -- one relu gate of MAG flipping on a 1e-7 difference; every
#  attention / feed-forward gradient is below 1e-3)
@pytest.mark.parametrize("cdt,L,tol_logit,tol_grad", [(torch.float32, 100, 1e-3, 5e-3), (torch.float32, 128, 1e-3, 1e-2),
                                                       (torch.bfloat16, 128, 5e-2, 1e-1), (torch.bfloat16, 72, 5e-2, 1e-1)])
def test_long_sequences_train_vs_oracle(cdt, L, tol_logit, tol_grad):
    """64 < L <= 128 (xlnet.py:104-146 builds the relative encoding for any klen; the reference takes any --max_seq_length): the
    strip-group kernels -- forward, query-side backward with its position window, key / position-side backward -- against the oracle
    in train mode (every dropout p = 0): logits, saved probabilities, every parameter gradient."""
    layers, B, nh = 2, 3, 12
    fp32 = cdt == torch.float32
    m = build(layers, cdt, p_mag=0.0, p=0.0).train()
    o = X.set_dropout(oracle(layers, p_mag=0.0), 0.0, 0.0).train()
    b = weights.synthetic_xlnet_batch(B, L, 47, 74, seed=90 + L)
    b["input_mask"][0, :] = 1                                    # one row without padding
    ids, vis, aco, mask, seg, lab = tb(b, DEV)
    i2, v2, a2, m2, s2, l2 = tb(b)
    out = m(ids, vis, aco, token_type_ids=seg, attention_mask=mask, output_attentions=True)
    logits, att = out[0], out[1]
    torch.nn.MSELoss()(logits.view(-1), lab.view(-1)).backward()
    lo = o(i2, v2, a2, m2, s2)[0]
    F.mse_loss(lo.view(-1), l2.view(-1)).backward()
    torch.cuda.synchronize()
    err = float((logits.detach().cpu() - lo.detach()).abs().max())
    perr = max(float((att[l].cpu() - lyr.rel_attn.last_probs.detach()).abs().max()) for l, lyr in enumerate(o.transformer.layer))
    print("xlnet L=%d (%s): logits %.2e, probabilities %.2e" % (L, cdt, err, perr))
    assert err <= tol_logit and perr <= (1e-5 if fp32 else 2e-2)
    _grad_report(m, o, tol_grad, frobenius=not fp32)
    # the fused single-call step runs at this length too (graph capture included)
    m.zero_grad()
    m.train_step(ids, vis, aco, mask, seg, lab, optimizer=None)
    torch.cuda.synchronize()


@pytest.mark.parametrize("B,L,ml,seed,cdt", [(4, 24, 24, 36, torch.float32), (3, 50, 40, 37, torch.float32), (3, 50, 40, 37, torch.bfloat16)])
def test_mems_match_reference_golden(golden, B, L, ml, seed, cdt):
    """f-4, mems (xlnet.py:81-91, 244-245, 276-293, 317-323, 363-385): segment 1 with use_cache and config.mem_len -> new_mems; segment 2
    consumes them (keys / values over cat([mem, h]): klen = 48, and 90 -- above the L = 64 kernel boundary) and caches again.
    Logits of both segments and samples of the cached memories against the values the REFERENCE's own xlnet.py produced
    (tests/golden/g6_xlnet.npz, oracle/make_golden.py), and -- attentions [B, n_head, qlen, klen], hidden states -- against the
    oracle run live.  fp32 <= 1e-3 (north_star); bf16 logits <= 5e-2."""
    fp32 = cdt == torch.float32
    g = golden["g6_xlnet"]
    tag = "B%d_L%d_M%d_seed%d" % (B, L, ml, seed)
    m = build(cdt=cdt, mem_len=ml).eval()
    o = oracle().eval()
    b1, b2 = weights.synthetic_xlnet_batch(B, L, 47, 74, seed=seed), weights.synthetic_xlnet_batch(B, L, 47, 74, seed=seed + 100)
    i1, v1, a1, m1, s1, _ = tb(b1, DEV)
    i2, v2, a2, m2, s2, _ = tb(b2, DEV)
    with torch.no_grad():
        r1 = m(i1, v1, a1, token_type_ids=s1, attention_mask=m1, use_cache=True)
        assert len(r1) == 2 and len(r1[1]) == 12 and tuple(r1[1][0].shape) == (min(ml, L), B, 768)
        r2 = m(i2, v2, a2, token_type_ids=s2, attention_mask=m2, use_cache=True, mems=list(r1[1]), output_attentions=True,
               output_hidden_states=True)
        c2 = tb(b2)
        mems_cpu = [t.cpu() for t in r1[1]]
        lo = o(c2[0], c2[1], c2[2], c2[3], c2[4], mems=mems_cpu, mem_len=ml)[0]
    tol = 1e-3 if fp32 else 5e-2
    e1 = float(np.abs(r1[0].cpu().numpy() - g["mems/logits_seg1/" + tag]).max())
    e2 = float(np.abs(r2[0].cpu().numpy() - g["mems/logits_seg2/" + tag]).max())
    eo = float((r2[0].cpu() - lo).abs().max())
    print("mems %s %s: logits vs reference golden seg 1 %.2e, seg 2 %.2e (vs the oracle fed OUR memories %.2e)" % (tag, cdt, e1, e2, eo))
    assert e1 <= tol and e2 <= tol and eo <= tol
    assert len(r2) == 4 and tuple(r2[1][0].shape) == (min(ml, 2 * L), B, 768)
    worst = 0.0
    for seg_name, mems in (("seg1", r1[1]), ("seg2", r2[1])):
        for i in (0, 1, 2, 11):
            ref = g["mems/new_mems_%s/%s/layer%d" % (seg_name, tag, i)]
            got = weights.strided_sample(mems[i].float().cpu().numpy(), 64)
            worst = max(worst, float(np.abs(got - ref).max()) / max(float(np.abs(ref).max()), 1e-6))
    print("new_mems samples vs reference golden: worst relative error %.2e" % worst)
    assert worst <= (1e-3 if fp32 else 3e-2)
    # hidden states [B, qlen, d] x 13 and attentions [B, n_head, qlen, klen] x 12 of the segment that consumed memories
    assert len(r2[2]) == 13 and tuple(r2[2][0].shape) == (B, L, 768)
    assert len(r2[3]) == 12 and tuple(r2[3][0].shape) == (B, 12, L, min(ml, L) + L)
    perr = max(float((r2[3][l].cpu() - lyr.rel_attn.last_probs).abs().max()) for l, lyr in enumerate(o.transformer.layer))
    print("attention probabilities over klen = %d keys vs the oracle: %.2e" % (min(ml, L) + L, perr))
    assert perr <= (1e-5 if fp32 else 2e-2)


@pytest.mark.parametrize("cdt,L,ml,tol", [(torch.float32, 24, 16, 5e-3), (torch.float32, 50, 40, 5e-3),
                                          (torch.bfloat16, 24, 16, 5e-2)])     # bf16: relative Frobenius; MAG's gated tensors 2e-1 (measured 1.0e-1: relu flips at T = 72)
def test_mems_training_gradients_vs_oracle(cdt, L, ml, tol):
    """Training WITH cached memories (xlnet.py:81-91, 374-385): keys / values of every layer over cat([mems[l], h]) with the memories
    detached.  The engine replaces the first mlen rows of every layer's input in the forward and clears their gradient at every seam
    in the backward; the k / v weights still receive the memory rows' share (the reference's einsum over cat does).  Train mode, p = 0:
    loss and every parameter gradient vs the oracle (klen 40: one strip group; klen 90: above the L = 64 kernel boundary)."""
    layers, B = 2, 3
    fp32 = cdt == torch.float32
    m = build(layers, cdt, p_mag=0.0, p=0.0).train()
    o = X.set_dropout(oracle(layers, p_mag=0.0), 0.0, 0.0).train()
    b = weights.synthetic_xlnet_batch(B, L, 47, 74, seed=91)
    g = torch.Generator().manual_seed(5)
    mems = [torch.randn(ml, B, 768, generator=g) * 0.5 for _ in range(layers)]
    ids, vis, aco, mask, seg, lab = tb(b, DEV)
    logits = m(ids, vis, aco, token_type_ids=seg, attention_mask=mask, mems=[t.to(DEV) for t in mems], labels=None)[0]
    torch.nn.MSELoss()(logits.view(-1), lab.view(-1)).backward()
    i2, v2, a2, m2, s2, l2 = tb(b)
    lo = o(i2, v2, a2, m2, s2, mems=mems)[0]
    F.mse_loss(lo.view(-1), l2.view(-1)).backward()
    torch.cuda.synchronize()
    err = float((logits.detach().cpu() - lo.detach()).abs().max())
    print("mems training (%s, klen %d): logits %.2e" % (cdt, ml + L, err))
    assert err <= (1e-3 if fp32 else 5e-2)
    _grad_report(m, o, tol, frobenius=not fp32, loose=() if fp32 else LOOSE_BF16, tol_loose=2e-1, show=3)
    # and the plain pass afterwards is unaffected (the engine's mems pointer is per pass)
    m.zero_grad()
    m.train_step(ids, vis, aco, mask, seg, lab, optimizer=None)
    torch.cuda.synchronize()


def test_mems_argument_checks():
    m = build(layers=2, mem_len=16)
    ids, vis, aco, mask, seg, _ = tb(weights.synthetic_xlnet_batch(2, 24, 47, 74, seed=3), DEV)
    mems = [torch.zeros(16, 2, 768) for _ in range(2)]
    with pytest.raises(NotImplementedError):                 # the differentiable BASE model does not take mems with autograd on
        m.transformer(ids, vis, aco, token_type_ids=seg, attention_mask=mask, mems=mems)
    m.eval()
    with torch.no_grad():
        with pytest.raises(ValueError):
            m(ids, vis, aco, token_type_ids=seg, attention_mask=mask, mems=mems[:1])
        with pytest.raises(NotImplementedError):             # klen = 120 + 24 > 128
            m(ids, vis, aco, token_type_ids=seg, attention_mask=mask, mems=[torch.zeros(120, 2, 768) for _ in range(2)])
        out = m(ids, vis, aco, token_type_ids=seg, attention_mask=mask, mems=mems, use_cache=False)
        assert len(out) == 1
    # the training step is unaffected by a memory pass before it (the engine's mems pointer is per pass)
    m.train()
    m.train_step(ids, vis, aco, mask, seg, torch.zeros(2, device=DEV), optimizer=None)
    torch.cuda.synchronize()


def test_too_long_sequence_is_rejected():
    m = build(layers=2).eval()
    ids, vis, aco, mask, seg, _ = tb(weights.synthetic_xlnet_batch(2, 130, 47, 74, seed=5), DEV)
    with pytest.raises(Exception):
        m(ids, vis, aco, token_type_ids=seg, attention_mask=mask)


def test_driver_epoch_xlnet_runs_and_learns():
    """bundled driver with --model xlnet-base-cased on synthetic prepare_xlnet_input-shaped data (bf16, dropout on)"""
    from bert_multimodal_transformer_amd import multimodal_driver as D
    D.args = D.parse_args(["--synthetic", "192", "--n_epochs", "1", "--seed", "5", "--train_batch_size", "48",
                           "--learning_rate", "5e-5", "--model", "xlnet-base-cased"])
    D.set_random_seed(D.args.seed)
    tr, dev, te, nsteps = D.set_up_data_loader()
    model = MAG_XLNetForSequenceClassification(XLNetConfig(n_layer=2), MultimodalConfig(1.0, 0.5), compute_dtype=torch.bfloat16)
    opt = AdamW(D.optimizer_grouped_parameters(model), lr=D.args.learning_rate)
    sch = get_linear_schedule_with_warmup(opt, num_warmup_steps=0, num_training_steps=1000)
    losses = [D.train_epoch(model, tr, opt, sch) for _ in range(4)]
    print("xlnet epoch losses", losses)
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]
    vl = D.eval_epoch(model, dev, opt)
    acc, mae, corr, f1 = D.test_score_model(model, te)
    assert np.isfinite(vl) and 0.0 <= acc <= 1.0 and np.isfinite(mae)
    D.args.reference_loop = True
    assert np.isfinite(D.train_epoch(model, tr, opt, sch))


def _xl_trajectory(mode, cdt=torch.float32, nsteps=4, shapes=((5, 40), (5, 40), (3, 24), (5, 40))):
    """nsteps optimizer steps (dropout ON) through model.train_step; mode False = passes driven from Python, True = step prologue +
    replayed hipGraph (mb_xlnet_train_step), "launches" = the same single call launching the kernels one by one"""
    from bert_multimodal_transformer_amd.multimodal_driver import optimizer_grouped_parameters
    torch.manual_seed(77)
    m = build(layers=3, cdt=cdt).train()
    opt = AdamW(optimizer_grouped_parameters(m), lr=1e-3)
    sch = get_linear_schedule_with_warmup(opt, num_warmup_steps=1.0, num_training_steps=10)
    losses = []
    with m.stream_scope():
        for s in range(nsteps):
            B, L = shapes[s % len(shapes)]
            ids, vis, aco, mask, seg, lab = tb(weights.synthetic_xlnet_batch(B, L, 47, 74, seed=90 + s), DEV)
            m.train_step(ids, vis, aco, mask, seg, lab, optimizer=opt, graph=mode)
            losses.append(m._core.loss_buf[0].clone())
            sch.step()
    stats = m._core.graph_stats()
    torch.cuda.synchronize()
    return dict(p=m.flat_params.clone(), g=m.flat_grads.clone(), losses=torch.stack(losses).cpu(), stats=stats,
                frozen=m.transformer.mask_emb.detach().clone())


def test_xlnet_single_call_step_equals_python_driven_passes():
    """mb_xlnet_train_step (step prologue + replayed hipGraph, or the same call launching kernel by kernel) ends every step where
    the passes driven from Python end it: same dropout masks, losses, parameters; gradients cleared; the frozen mask_emb slot
    untouched.  Shapes change inside the run (two graphs, replays)."""
    ref = _xl_trajectory(False)
    noise = float((ref["p"] - _xl_trajectory(False)["p"]).abs().max())
    assert ref["stats"] == (0, 0)
    for name, mode in (("launches", "launches"), ("graph", True)):
        run = _xl_trajectory(mode)
        dp = float((run["p"] - ref["p"]).abs().max())
        dl = float((run["losses"] - ref["losses"]).abs().max())
        print("xlnet single call (%s) vs python-driven: |dparam| %.3e |dloss| %.3e (run-to-run %.3e) graphs %s" % (name, dp, dl, noise, run["stats"]))
        assert dp <= 1e-5 + 10 * noise and dl <= 1e-5
        assert float(run["g"].abs().max()) == 0.0
        assert torch.equal(run["frozen"], ref["frozen"])
        assert run["stats"] == ((2, 4) if mode is True else (0, 0))


def test_xlnet_adamw_riders_change_nothing(monkeypatch):
    """MB_ADAMW_RIDE (csrc/kernels.h AdamRide) in the MAG-XLNet engine: the ffn1 / ffn2 / out dgrad launches of layer l carry the HF-AdamW update
    of layers l+1.. as rider workgroups and the sweep at the end of the step skips what they did.  Same arithmetic per element, so in
    deterministic mode four bf16 steps (dropout on, schedule moving, two shapes -- one wide enough for the 128 x 128 ffn2 host: T = 1,200) end
    with the SAME BITS in the parameters; gradients read as zeros, the frozen mask_emb slot is untouched."""
    monkeypatch.setenv("MB_DETERMINISTIC", "1")
    shapes = ((24, 50), (24, 50), (5, 40), (24, 50))
    monkeypatch.setenv("MB_ADAMW_RIDE", "0")
    ref = _xl_trajectory(True, torch.bfloat16, shapes=shapes)
    monkeypatch.setenv("MB_ADAMW_RIDE", "1")
    ride = _xl_trajectory(True, torch.bfloat16, shapes=shapes)
    assert torch.equal(ride["p"], ref["p"]), "parameters differ with the update riding in the backward launches: max %.3e" % float((ride["p"] - ref["p"]).abs().max())
    assert torch.allclose(ride["losses"], ref["losses"], rtol=1e-6, atol=0.0) and float(ride["g"].abs().max()) == 0.0      # (the batch mean is a float atomic sum)
    assert torch.equal(ride["frozen"], ref["frozen"]) and ride["stats"] == ref["stats"]


def test_input_mask_and_perm_mask_match_reference_golden_fp32(golden):
    """xlnet.py:258-296: `input_mask` (1 = padding) in place of attention_mask gives the same logits, and a `perm_mask` [B, L, L]
    (1 = query i may not attend to key j; the i == j exemption of non_tgt_mask stays) moves them to what the REFERENCE's own
    xlnet.py computed (fixture entries written by oracle/make_golden.py from the reference stack); passing both masks asserts
    like the reference does."""
    g = golden["g6_xlnet"]
    m = build().eval()
    ids, vis, aco, mask, seg, lab = tb(weights.synthetic_xlnet_batch(4, 50, 47, 74, seed=31), DEV)
    perm = torch.from_numpy(g["perm_mask/B4_L50_rs77"].astype(np.float32)).to(DEV)
    with torch.no_grad():
        a = m(ids, vis, aco, token_type_ids=seg, input_mask=1.0 - mask.float())[0].cpu().numpy()
        b = m(ids, vis, aco, token_type_ids=seg, attention_mask=mask, perm_mask=perm)[0].cpu().numpy()
        c = m(ids, vis, aco, token_type_ids=seg, attention_mask=mask)[0].cpu().numpy()          # and a plain forward afterwards is unaffected
    e_im = float(np.abs(a - g["logits_input_mask/B4_L50_seed31"]).max())
    e_pm = float(np.abs(b - g["logits_perm_mask/B4_L50_seed31"]).max())
    e_pl = float(np.abs(c - g["logits/B4_L50_seed31"]).max())
    moved = float(np.abs(g["logits_perm_mask/B4_L50_seed31"] - g["logits/B4_L50_seed31"]).max())
    print("xlnet input_mask logits err %.2e, perm_mask logits err %.2e (the mask moves them by %.2e), plain %.2e" % (e_im, e_pm, moved, e_pl))
    assert e_im <= 1e-3 and e_pm <= 1e-3 and e_pl <= 1e-3 and moved > 1e-2
    with pytest.raises(AssertionError):
        m(ids, vis, aco, token_type_ids=seg, attention_mask=mask, input_mask=1.0 - mask.float())


@pytest.mark.parametrize("B,L,M,seed,cdt,tol_g,tol", [(4, 50, 5, 41, torch.float32, 1e-3, 1e-3), (2, 100, 9, 42, torch.float32, 1e-3, 1e-3),
                                                      (4, 50, 5, 41, torch.bfloat16, 1e-1, 5e-2)])
def test_query_stream_matches_reference_golden(golden, B, L, M, seed, cdt, tol_g, tol):
    """f-4, target_mapping (xlnet.py:238-240, 306-313, 374-399): the query stream g -- mask_emb on M target rows per sample, per layer the
    layer's q projection, the map onto the L positions, attention over the content stream's keys / values under the perm_mask WITHOUT
    the self exemption, the map back, post_attention + feed-forward -- against what the REFERENCE's own xlnet.py returned for the same
    inputs (oracle/make_golden.py gen_xlnet): output_g [B, M, 768] (what MAG_XLNetModel returns first) and the classifier's logits on
    it.  fp32: the north-star 1e-3; bf16 activations: 1e-1 on LayerNorm outputs of magnitude ~3.8 (bf16 spacing there 1.6e-2, twelve
    layers deep), logits 5e-2 as the other bf16 eval tests."""
    g = golden["g6_xlnet"]
    tag = "B%d_L%d_M%d_seed%d" % (B, L, M, seed)
    m = build(cdt=cdt).eval()
    ids, vis, aco, mask, seg, lab = tb(weights.synthetic_xlnet_batch(B, L, 47, 74, seed=seed), DEV)
    tm = torch.from_numpy(g["target_mapping/tm/" + tag].astype(np.float32)).to(DEV)
    pm = torch.from_numpy(g["target_mapping/perm/" + tag].astype(np.float32)).to(DEV)
    with torch.no_grad():
        plain = m(ids, vis, aco, token_type_ids=seg, attention_mask=mask, perm_mask=pm)[0].cpu().numpy()
        r = m.transformer(ids, vis, aco, token_type_ids=seg, attention_mask=mask, perm_mask=pm, target_mapping=tm, output_hidden_states=True)
        logits = m(ids, vis, aco, token_type_ids=seg, attention_mask=mask, perm_mask=pm, target_mapping=tm)[0].cpu().numpy()
        again = m(ids, vis, aco, token_type_ids=seg, attention_mask=mask, perm_mask=pm)[0].cpu().numpy()
    out_g = r[0].cpu().numpy()
    ref_g, ref_l = g["target_mapping/output_g/" + tag], g["target_mapping/logits/" + tag]
    e_g, e_l = float(np.abs(out_g - ref_g).max()), float(np.abs(logits - ref_l).max())
    print("query stream %s %s: output_g err %.2e (|g| max %.2f), logits err %.2e; the g head differs from the h head by %.2e"
          % (tag, cdt, e_g, float(np.abs(ref_g).max()), e_l, float(np.abs(ref_l - plain).max())))
    assert out_g.shape == (B, M, 768) and e_g <= tol_g and e_l <= tol
    assert np.array_equal(plain, again)                       # the post-pass leaves nothing behind in the engine
    # hidden_states: (h_0, g_0, h_1, g_1, ...) (xlnet.py:412-416); g_0 is mask_emb on every row
    hs = r[1]
    assert len(hs) == 2 * 13 and tuple(hs[0].shape) == (B, L, 768) and tuple(hs[1].shape) == (B, M, 768)
    me = m.transformer.mask_emb.detach().float().view(1, 1, 768)
    assert float((hs[1] - me.to(hs[1].dtype).float()).abs().max()) <= (0.0 if cdt == torch.float32 else 1e-2)
    assert float((hs[-1] - r[0]).abs().max()) == 0.0


def test_query_stream_general_mapping_and_argument_checks():
    """A target_mapping with fractional weights (two positions per target) against the oracle run live; the modes the stream is not
    built for raise."""
    B, L, M, layers = 3, 40, 4, 3
    m = build(layers=layers).eval()
    o = oracle(layers=layers).eval()
    b = weights.synthetic_xlnet_batch(B, L, 47, 74, seed=51)
    ids, vis, aco, mask, seg, lab = tb(b, DEV)
    rs = np.random.RandomState(5)
    tm = np.zeros((B, M, L), np.float32)
    for bb in range(B):
        for mm in range(M):
            i, j = rs.choice(np.arange(L - 12, L), size=2, replace=False)
            tm[bb, mm, i], tm[bb, mm, j] = 0.75, 0.25
    pm = (rs.rand(B, L, L) < 0.2).astype(np.float32)
    tm_t, pm_t = torch.from_numpy(tm), torch.from_numpy(pm)
    hm = torch.from_numpy(rs.rand(layers, 12).astype(np.float32))
    with torch.no_grad():
        got = m.transformer(ids, vis, aco, token_type_ids=seg, attention_mask=mask, perm_mask=pm_t.to(DEV), target_mapping=tm_t.to(DEV),
                            head_mask=hm.to(DEV))[0].cpu()
        c = tb(b)
        want = o.transformer(c[0], c[1], c[2], c[3], c[4], perm_mask=pm_t, target_mapping=tm_t, head_mask=hm)
    err = float((got - want).abs().max())
    print("query stream, fractional mapping + head_mask + random perm_mask: output_g err %.2e (|g| max %.2f)" % (err, float(want.abs().max())))
    assert err <= 1e-3
    with pytest.raises(NotImplementedError):                 # autograd on
        m(ids, vis, aco, token_type_ids=seg, attention_mask=mask, target_mapping=tm_t.to(DEV))
    with torch.no_grad():
        with pytest.raises(NotImplementedError):
            m(ids, vis, aco, token_type_ids=seg, attention_mask=mask, target_mapping=tm_t.to(DEV), output_attentions=True)
        with pytest.raises(NotImplementedError):
            m(ids, vis, aco, token_type_ids=seg, attention_mask=mask, target_mapping=tm_t.to(DEV),
              mems=[torch.zeros(8, B, 768) for _ in range(layers)])
        with pytest.raises(ValueError):
            m(ids, vis, aco, token_type_ids=seg, attention_mask=mask, target_mapping=tm_t[:, :, :-1].to(DEV))
        m.train()
        with pytest.raises(NotImplementedError):
            m(ids, vis, aco, token_type_ids=seg, attention_mask=mask, target_mapping=tm_t.to(DEV))


@pytest.mark.parametrize("cdt", [torch.float32, torch.bfloat16])
def test_perm_mask_gradients_vs_oracle(cdt):
    """train mode (every dropout p = 0), a random perm_mask plus ragged padding, L = 72 (two strip groups): logits and every
    parameter gradient against the oracle -- the backward works from the saved probabilities, which are exact zeros wherever the
    forward masked a score -- then the single-call step still runs (the mask is an argument of explicit forwards only)."""
    layers, B, L = 2, 3, 72
    fp32 = cdt == torch.float32
    m = build(layers, cdt, p_mag=0.0, p=0.0).train()
    o = X.set_dropout(oracle(layers, p_mag=0.0), 0.0, 0.0).train()
    b = weights.synthetic_xlnet_batch(B, L, 47, 74, seed=58)
    ids, vis, aco, mask, seg, lab = tb(b, DEV)
    perm = torch.from_numpy((np.random.RandomState(3).rand(B, L, L) < 0.4).astype(np.float32))
    logits = m(ids, vis, aco, token_type_ids=seg, attention_mask=mask, perm_mask=perm.to(DEV))[0]
    torch.nn.MSELoss()(logits.view(-1), lab.view(-1)).backward()
    i2, v2, a2, m2, s2, l2 = tb(b)
    lo = o(i2, v2, a2, m2, s2, perm_mask=perm)[0]
    torch.nn.functional.mse_loss(lo.view(-1), l2.view(-1)).backward()
    torch.cuda.synchronize()
    err = float((logits.detach().cpu() - lo.detach()).abs().max())
    print("xlnet perm_mask (%s): logits %.2e" % (cdt, err))
    assert err <= (1e-3 if fp32 else 5e-2)
    _grad_report(m, o, 5e-3 if fp32 else 1e-1, frobenius=not fp32)
    m.zero_grad()
    m.train_step(ids, vis, aco, mask, seg, lab, optimizer=None)
    torch.cuda.synchronize()


_XL_DET_WORKER = r"""
import os, sys, torch
sys.path.insert(0, os.environ["REPO_ROOT"]); sys.path.insert(0, os.path.join(os.environ["REPO_ROOT"], "tests"))
from test_xlnet_gpu import build, tb, weights, DEV
from bert_multimodal_transformer_amd import AdamW, get_linear_schedule_with_warmup
from bert_multimodal_transformer_amd.multimodal_driver import optimizer_grouped_parameters
torch.manual_seed(11)
m = build(layers=3, cdt=torch.bfloat16).train()
opt = AdamW(optimizer_grouped_parameters(m), lr=1e-3)
sch = get_linear_schedule_with_warmup(opt, 0, 100)
mode = os.environ["MODE"]
with m.stream_scope():
    for s in range(4):
        ids, vis, aco, mask, seg, lab = tb(weights.synthetic_xlnet_batch(6, 50, 47, 74, seed=70 + s), DEV)
        m.train_step(ids, vis, aco, mask, seg, lab, optimizer=opt, graph={"graph": True, "launches": "launches", "python": False}[mode])
        sch.step()
torch.cuda.synchronize()
torch.save(m.flat_params.cpu(), os.environ["OUT"])
print("OK")
"""


def test_deterministic_mode_bf16_runs_are_bit_identical(tmp_path):
    """MB_DETERMINISTIC=1 for MAG-XLNet ("two runs bit-identical" in place of the absent sanitizers): every
    multi-writer gradient sum of the engine -- r_w / r_r / r_s bias and seg_embed column sums of the attention backward, the
    word-embedding scatter, LayerNorm / bias slabs, MAG's gate sums, the head -- goes through the 64-bit fixed-point shadow, so
    two bf16 training runs (dropout on, 4 steps) end with the SAME bits, and so do the replayed graph, the launch-by-launch call
    and the passes driven from Python."""
    import subprocess, sys
    script = tmp_path / "xd.py"
    script.write_text(_XL_DET_WORKER)
    outs = {}
    for tag, mode in (("a", "graph"), ("b", "graph"), ("c", "launches"), ("d", "python")):
        out = str(tmp_path / ("p_" + tag))
        env = dict(os.environ, REPO_ROOT=ROOT, OUT=out, MODE=mode, MB_DETERMINISTIC="1")
        p = subprocess.run([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
        assert p.returncode == 0, p.stdout.decode()[-3000:]
        outs[tag] = torch.load(out)
    assert torch.equal(outs["a"], outs["b"]), float((outs["a"] - outs["b"]).abs().max())
    assert torch.equal(outs["a"], outs["c"]), float((outs["a"] - outs["c"]).abs().max())
    assert torch.equal(outs["a"], outs["d"]), float((outs["a"] - outs["d"]).abs().max())
