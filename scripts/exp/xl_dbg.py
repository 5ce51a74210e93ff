import os, sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import test_xlnet_gpu as TX
from test_xlnet_gpu import *
def run(L, cdt=torch.float32):
    layers, B, nh, H, DI = 2, 3, 12, 768, 3072
    torch.manual_seed(99)
    m = build(layers, cdt).train()
    o = oracle(layers).train()
    core = m._core
    b = weights.synthetic_xlnet_batch(B, L, 47, 74, seed=41)
    ids, vis, aco, mask, seg, lab = tb(b, DEV)
    out = m(ids, vis, aco, token_type_ids=seg, attention_mask=mask, labels=None)
    logits = out[0]
    torch.nn.MSELoss()(logits.view(-1), lab.view(-1)).backward()
    seed, step = core.seed, core.step
    mult = lambda site, p, n: torch.from_numpy(rng.keep_mult(n, rng.make_key(seed, step, site, p)))
    blx = lambda site, p, Xd: mult(site, p, B * L * Xd).view(B, L, Xd).permute(1, 0, 2)
    S = TX._SeqReplay
    o.transformer.dropout = S([blx(rng.XS_EMB, 0.1, H), mult(rng.XS_POS, 0.1, 2 * L * B * H).view(2 * L, B, H), blx(rng.XS_FINAL, 0.1, H)])
    o.transformer.MAG.dropout = S([blx(rng.XS_MAG, 0.5, H)])
    o.sequence_summary.last_dropout = S([mult(rng.XS_HEAD, 0.1, B * H).view(B, H)])
    for l, lyr in enumerate(o.transformer.layer):
        s0 = rng.XS_LAYER0 + 8 * l
        lyr.rel_attn.dropout = S([mult(s0 + 0, 0.1, B * nh * L * L).view(B, nh, L, L), blx(s0 + 1, 0.1, H)])
        lyr.ff.dropout = S([blx(s0 + 2, 0.1, DI), blx(s0 + 3, 0.1, H)])
    i2, v2, a2, m2, s2, l2 = tb(b)
    lo = o(i2, v2, a2, m2, s2)[0]
    torch.nn.functional.mse_loss(lo.view(-1), l2.view(-1)).backward()
    torch.cuda.synchronize()
    og = {n: p.grad for n, p in o.named_parameters() if p.grad is not None}
    gmax = max(float(g.abs().max()) for g in og.values())
    rows = []
    for n, p in m.named_parameters():
        if n in og:
            g, r = p.grad.detach().cpu(), og[n]
            rows.append((float((g - r).abs().max()) / max(float(r.abs().max()), 1e-3 * gmax), n))
    rows.sort(reverse=True)
    print("L=%d %s lib=%s" % (L, cdt, os.environ.get("MB_LIB_DIR", "tree")))
    for e, n in rows[:12]:
        print("   %.3e %s" % (e, n))
for L in (24, 50):
    run(L)
