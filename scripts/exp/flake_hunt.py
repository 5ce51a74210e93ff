"""Hunt for the once-seen wrong gradient (NOTES_NEXT_ROUND idea 5 / VERDICT r4 weak #1): tests/test_xlnet_gpu.py::
test_train_mode_dropout_mask_replay[fp32, L=24] once gave 7.9e-2 at MAG.W_ha.weight with exact logits and probabilities.

The case (2-layer MAG-XLNet, B=3, L=24 -> T=72 tokens, dropout on at every site, autograd route: forward + backward) is repeated
N times per configuration on a FRESH model / engine each time with the dropout counter pinned, so every repetition must reproduce the
same flat gradient: the first repetition is checked against the CPU oracle (mask replay, as the test does), the others against the
first on the device (bit-equal under MB_DETERMINISTIC=1, <= 2e-5 of the largest gradient otherwise; the flake was 7.9e-2).
Configurations: caller on the NULL stream | on a private stream; each alone | with a second engine (2-layer MAG-BERT training
steps on another stream) running underneath; every repetition preceded by allocator churn whose blocks are POISONED with NaN before
they go back to torch's cache (an uninitialised read of recycled memory then shows up as NaN instead of a plausible number).
On the first miss: per-tensor table against the reference gradient + the location of the differing elements, then continue.

    python scripts/exp/flake_hunt.py [--n 300] [--L 24] [--dtype fp32|bf16]        (MB_DETERMINISTIC is read from the environment)

--vary N: the second hypothesis.  The test never seeds torch, so the engine's dropout seed (torch.initial_seed() at model creation)
-- and with it every mask and every activation behind the first dropout -- differs from one pytest run to the next; MAG's two
relu gates and its min(threshold, 1) clamp have discontinuous derivatives, so a pre-activation that lands within fp32 rounding of
zero can take different sides on the CPU and on the GPU: identical logits, one token's contribution to dW_hv / dW_ha flipped.  N
repetitions with a DIFFERENT dropout draw each (one model, the counter advances), each against the CPU oracle with the masks
replayed; per repetition the smallest |relu pre-activation| of the oracle's MAG is recorded; on a miss the oracle is re-run in
float64 and the GPU and the fp32 oracle are both compared with it.
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch                                                        # noqa: E402
import test_xlnet_gpu as TX                                          # noqa: E402
import test_model_gpu as TB                                          # noqa: E402
from bert_multimodal_transformer_amd import rng                      # noqa: E402
from oracle import weights                                           # noqa: E402

DEV = "cuda:0"


oracle_grads = TX.oracle_replay_grads          # (shared with the regression test of the finding)


def vary(a, cdt):
    """a different dropout draw per repetition, GPU vs CPU oracle each time (see the module docstring)"""
    layers, B, L = 2, 3, a.L
    b = weights.synthetic_xlnet_batch(B, L, 47, 74, seed=41)
    batch = TX.tb(b, DEV)
    ids, vis, aco, mask, seg, lab = batch
    torch.manual_seed(a.seed)
    m = TX.build(layers, cdt).train()
    tol = 5e-3 if cdt == torch.float32 else 1e-1
    misses, margins, t0 = 0, [], time.time()
    for rep in range(a.vary):
        m.zero_grad()
        out = m(ids, vis, aco, token_type_ids=seg, attention_mask=mask, labels=None)
        torch.nn.MSELoss()(out[0].view(-1), lab.view(-1)).backward()
        torch.cuda.synchronize()
        seed, step = m._core.seed, m._core.step
        probe = []
        og, lo = oracle_grads(layers, B, L, seed, step, b, probe=probe)
        clamp_margin = float(probe[0].abs().min())          # (the MAG pre-hook fires first: probe = [threshold - 1, W_hv, W_ha])
        probe = probe[1:]
        margin = min(float(x.abs().min()) for x in probe)
        margins.append(margin)
        gmax = max(float(g.abs().max()) for g in og.values())
        rows = sorted(((float((p.grad.detach().cpu() - og[n]).abs().max()) / max(float(og[n].abs().max()), 1e-3 * gmax), n)
                       for n, p in m.named_parameters() if n in og), reverse=True)
        if rows[0][0] > tol:
            misses += 1
            lerr = float((out[0].detach().cpu() - lo).abs().max())
            print("  MISS rep=%d (seed %d, step %d): worst gradient %.3e at %s, logits %.2e, smallest |relu pre-activation| %.3e, smallest |clamp "
                  "threshold - 1| %.3e" % (rep, seed, step, rows[0][0], rows[0][1], lerr, margin, clamp_margin))
            for e_, n_ in rows[:5]:
                print("      %.3e  %s" % (e_, n_))
            probe64 = []
            o64, _ = oracle_grads(layers, B, L, seed, step, b, double=True, probe=probe64)
            probe64 = probe64[1:]
            for nm, x32, x64 in zip(("W_hv", "W_ha"), probe, probe64):
                flip = ((x32 > 0) != (x64 > 0)).nonzero()
                print("      relu gate %s: %d of %d pre-activations have different signs in the fp32 and the float64 oracle%s" %
                      (nm, flip.shape[0], x32.numel(), "".join("; [%s] fp32 %.3e float64 %.3e" % (",".join(str(int(i)) for i in ix), float(x32[tuple(ix)]), float(x64[tuple(ix)]))
                                                                for ix in flip[:3])))
            g64max = max(float(g.abs().max()) for g in o64.values())
            for n_ in [r[1] for r in rows[:3]]:
                ref = o64[n_]
                den = max(float(ref.abs().max()), 1e-3 * g64max)
                p_ = dict(m.named_parameters())[n_]
                print("      %s vs the float64 oracle: GPU %.3e, fp32 CPU oracle %.3e" %
                      (n_, float((p_.grad.detach().cpu().double() - ref).abs().max()) / den, float((og[n_].double() - ref).abs().max()) / den))
            # the decisive check: invert the sign of ONE gate pre-activation in the float64 oracle -- each of those within 1e-5 of zero
            # in turn -- and compare again: if the GPU merely took the other side of one relu, one of them reproduces its gradient
            cands = []
            for k in range(2):
                flat = probe64[k].abs().flatten()
                for j in torch.topk(flat, 12, largest=False).indices.tolist():
                    if float(flat[j]) <= 1e-5:
                        cands.append((float(flat[j]), k, [int(i) for i in torch.unravel_index(torch.tensor(j), probe64[k].shape)]))
            cands = sorted(cands)[:16]
            worst = lambda ref_: max(float((p_.grad.detach().cpu().double() - ref_[n_]).abs().max()) / max(float(ref_[n_].abs().max()), 1e-3 * g64max)
                                     for n_, p_ in m.named_parameters() if n_ in ref_)
            best = None
            for mag, k, idx in sorted(cands):
                of, _ = oracle_grads(layers, B, L, seed, step, b, double=True, flip=(("W_hv", "W_ha")[k], idx))
                w_ = worst(of)
                if best is None or w_ < best[0]:
                    best = (w_, k, idx, float(probe64[k][tuple(idx)]))
            print("      every tensor, GPU vs float64 oracle: %.3e as is; %.3e with the sign of ONE pre-activation inverted: %s[%s] = %.3e" %
                  (worst(o64), best[0], ("W_hv", "W_ha")[best[1]], ",".join(map(str, best[2])), best[3]))
    ms = sorted(margins)
    print("vary: %d repetitions with different dropout draws (%s, L=%d), %d misses at tolerance %.0e; smallest |relu pre-activation| per "
          "repetition: min %.2e, median %.2e   (%.0f s)" % (a.vary, a.dtype, L, misses, tol, ms[0], ms[len(ms) // 2], time.time() - t0))
    return misses


def churn(rep):
    """allocate a handful of odd-sized blocks, fill them with NaN, free them: what the next model / workspace allocation recycles"""
    g = torch.Generator().manual_seed(1000 + rep)
    sizes = torch.randint(1 << 12, 48 << 20, (6,), generator=g).tolist()
    blocks = [torch.empty(int(s), dtype=torch.float32, device=DEV).fill_(float("nan")) for s in sizes]
    del blocks


def one(layers, B, L, cdt, sd_dev, batch):
    torch.manual_seed(99)
    m = TX.build(layers, cdt).train()
    ids, vis, aco, mask, seg, lab = batch
    out = m(ids, vis, aco, token_type_ids=seg, attention_mask=mask, labels=None)
    torch.nn.MSELoss()(out[0].view(-1), lab.view(-1)).backward()
    return m, out[0].detach()


def table(m, ref_flat, names_ref=None):
    flat = m.flat_grads.detach()
    rows = []
    gmax = float(ref_flat.abs().max())
    for name, off, numel, shape, decay in m._core.tensors:
        a, r = flat[off: off + numel], ref_flat[off: off + numel]
        d = (a - r).abs()
        bad = d != d
        e = float(torch.where(bad, torch.full_like(d, float("inf")), d).max()) if numel else 0.0
        rows.append((e / max(float(r.abs().max()), 1e-3 * gmax), name, shape, int((d > 1e-4 * max(float(r.abs().max()), 1e-3 * gmax)).sum() + bad.sum()), int(bad.sum())))
    rows.sort(key=lambda x: -x[0])
    for rel, name, shape, nbad, nnan in rows[:10]:
        print("      %.3e  %-60s %s  elements off: %d (NaN %d)" % (rel, name, tuple(shape), nbad, nnan))
    rel, name, shape, nbad, _ = rows[0]
    for nm, off, numel, shp, _d in m._core.tensors:
        if nm == name and len(shp) == 2:
            d = (flat[off: off + numel] - ref_flat[off: off + numel]).view(*shp)
            d = torch.where(d != d, torch.full_like(d, 1e30), d).abs()
            rws = (d.max(dim=1).values > 1e-4 * float(ref_flat[off: off + numel].abs().max())).nonzero().view(-1)
            cls = (d.max(dim=0).values > 1e-4 * float(ref_flat[off: off + numel].abs().max())).nonzero().view(-1)
            print("      %s: %d rows off (%s ..), %d columns off (%s ..)" % (name, rws.numel(), rws[:8].tolist(), cls.numel(), cls[:8].tolist()))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=300)
    ap.add_argument("--L", type=int, default=24)
    ap.add_argument("--dtype", default="fp32")
    ap.add_argument("--vary", type=int, default=0)
    ap.add_argument("--seed", type=int, default=12345)
    a = ap.parse_args()
    cdt = torch.float32 if a.dtype == "fp32" else torch.bfloat16
    if a.vary > 0:
        torch.cuda.set_device(0)
        vary(a, cdt)
        return 0
    det = os.environ.get("MB_DETERMINISTIC", "0") == "1"
    layers, B, L = 2, 3, a.L
    torch.cuda.set_device(0)
    b = weights.synthetic_xlnet_batch(B, L, 47, 74, seed=41)
    batch = TX.tb(b, DEV)
    # reference run + oracle check
    m, logits = one(layers, B, L, cdt, None, batch)
    torch.cuda.synchronize()
    og, lo = oracle_grads(layers, B, L, m._core.seed, m._core.step, b)
    gmax = max(float(g.abs().max()) for g in og.values())
    worst = max((float((p.grad.detach().cpu() - og[n]).abs().max()) / max(float(og[n].abs().max()), 1e-3 * gmax), n) for n, p in m.named_parameters() if n in og)
    print("reference repetition vs CPU oracle: logits %.2e, worst gradient %.3e at %s" % (float((logits.cpu() - lo).abs().max()), worst[0], worst[1]))
    tol_oracle = 5e-3 if cdt == torch.float32 else 1e-1
    assert worst[0] <= tol_oracle, "the reference repetition itself is off"
    ref = m.flat_grads.detach().clone()
    ref_logits = logits.clone()
    gref = float(ref.abs().max())
    tol = 0.0 if det else 2e-5 * gref
    del m
    # the second engine (MAG-BERT, its own stream)
    side = torch.cuda.Stream()
    mb = TB.build(47, 2, cdt).train()
    bb = TB.tb(weights.synthetic_bert_batch(8, 50, 47, 74, seed=7), DEV)
    own = torch.cuda.Stream()
    total = misses = 0
    t0 = time.time()
    for cfg in ("null", "null+engine2", "private", "private+engine2"):
        cm = 0
        for rep in range(a.n):
            churn(rep)
            if "engine2" in cfg:
                with torch.cuda.stream(side):
                    for _ in range(3):
                        mb.train_step(*bb, optimizer=None)
            if cfg.startswith("private"):
                own.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(own):
                    m, logits = one(layers, B, L, cdt, None, batch)
                torch.cuda.current_stream().wait_stream(own)
            else:
                m, logits = one(layers, B, L, cdt, None, batch)
            flat = m.flat_grads.detach()
            d = (flat - ref).abs()
            bad = bool((d != d).any()) or float(d.max()) > tol or not torch.equal(logits, ref_logits) and det
            total += 1
            if bad:
                misses += 1; cm += 1
                print("  MISS cfg=%s rep=%d: max |d| %.3e (%.3e of the largest gradient), logits equal: %s" %
                      (cfg, rep, float(d.max()), float(d.max()) / gref, bool(torch.equal(logits, ref_logits))))
                if cm <= 3:
                    table(m, ref)
            del m
        torch.cuda.synchronize()
        print("cfg %-16s: %d repetitions, %d misses   (%.0f s so far)" % (cfg, a.n, cm, time.time() - t0))
    print("flake hunt %s L=%d MB_DETERMINISTIC=%s: %d repetitions, %d misses (criterion: %s)" %
          (a.dtype, L, "1" if det else "0", total, misses, "bit-equal" if det else "<= 2e-5 of the largest gradient"))
    return 1 if misses else 0


if __name__ == "__main__":
    sys.exit(main())
