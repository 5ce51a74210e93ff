"""Round 6, VERDICT r5 item 5a: the one miss of scripts/exp/flake_hunt.py --vary (profiles/r05_flake_hunt.txt, draw 63 = seed 12345, step 64:
8.6e-3 at MAG.W_ha.weight) that no SINGLE inverted relu gate reproduced.  Row-wise diagnosis: row c of dW_ha is sum_t gate'(pre[t, c]) *
upstream[t, c] * x_t, so the rows where the GPU differs from the float64 oracle name the output channels whose gate state differs, and the
difference row is a signed sum of the inputs x_t of the tokens whose gate flipped.  For every such row the candidate tokens (smallest
|pre-activation| of that channel) are flipped in the float64 oracle -- singly, then as sets -- until the GPU gradient is reproduced.

    python scripts/exp/flake_draw63.py [--step 64] [--seed 12345]"""
import argparse
import itertools
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch                                                        # noqa: E402
import test_xlnet_gpu as TX                                          # noqa: E402
from oracle import weights                                           # noqa: E402

DEV = "cuda:0"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=12345)
    ap.add_argument("--step", type=int, default=64)
    a = ap.parse_args()
    layers, B, L = 2, 3, 24
    b = weights.synthetic_xlnet_batch(B, L, 47, 74, seed=41)
    ids, vis, aco, mask, seg, lab = TX.tb(b, DEV)
    torch.manual_seed(a.seed)
    m = TX.build(layers, torch.float32).train()
    m._core.step = a.step - 1
    out = m(ids, vis, aco, token_type_ids=seg, attention_mask=mask, labels=None)
    torch.nn.MSELoss()(out[0].view(-1), lab.view(-1)).backward()
    torch.cuda.synchronize()
    seed, step = m._core.seed, m._core.step
    print("draw: seed %d step %d" % (seed, step))
    gpu = {n: p.grad.detach().cpu().double() for n, p in m.named_parameters() if p.grad is not None}
    probe = []
    o64, lo = TX.oracle_replay_grads(layers, B, L, seed, step, b, double=True, probe=probe)
    probe32 = []
    o32, _ = TX.oracle_replay_grads(layers, B, L, seed, step, b, probe=probe32)
    pre = {"W_hv": probe[1], "W_ha": probe[2]}                     # [L, B, 768] (the oracle runs sequence-major)
    pre32 = {"W_hv": probe32[1], "W_ha": probe32[2]}
    print("logits GPU vs float64 oracle: %.2e" % float((out[0].detach().cpu().double() - lo).abs().max()))
    gmax = max(float(g.abs().max()) for g in o64.values())

    def worst(ref):
        rows = sorted(((float((gpu[n] - ref[n]).abs().max()) / max(float(ref[n].abs().max()), 1e-3 * gmax), n) for n in ref if n in gpu), reverse=True)
        return rows

    rows = worst(o64)
    print("as is: worst %.3e at %s ; then %.3e %s" % (rows[0] + rows[1]))
    flips = []
    for gate in ("W_hv", "W_ha"):
        name = "transformer.MAG.%s.weight" % gate
        D = gpu[name] - o64[name]                                    # [768, in]
        scale = float(o64[name].abs().max())
        rowerr = D.abs().max(dim=1).values / scale
        off = (rowerr > 1e-5).nonzero().view(-1).tolist()
        print("%s: %d of 768 rows differ from the float64 oracle by more than 1e-5 of the tensor's maximum: %s" % (gate, len(off), off[:12]))
        for c in off:
            col = pre[gate][:, :, c]                                 # [L, B]
            order = torch.argsort(col.abs().flatten())[:6].tolist()
            cands = [(int(j // B), int(j % B)) for j in order]
            print("   row %d (error %.3e): smallest |pre-activation| of this channel: %s" % (
                c, float(rowerr[c]), ", ".join("[l=%d,b=%d] f64 %.3e fp32 %.3e" % (l_, b_, float(col[l_, b_]), float(pre32[gate][l_, b_, c])) for l_, b_ in cands[:4])))
            # which subset of the four nearest-to-zero tokens, flipped, reproduces this ROW of the GPU gradient?
            best = None
            near = [(l_, b_) for l_, b_ in cands[:4] if abs(float(col[l_, b_])) <= 1e-4]
            for r in range(1, len(near) + 1):
                for sub in itertools.combinations(near, r):
                    fl = [(gate, [l_, b_, c]) for l_, b_ in sub]
                    of, _ = TX.oracle_replay_grads(layers, B, L, seed, step, b, double=True, flip=fl)
                    e = float((gpu[name][c] - of[name][c]).abs().max()) / scale
                    if best is None or e < best[0]:
                        best = (e, fl)
            if near:
                # how much of the "gate open" contribution does the GPU row carry?  (closed: 0, open: 1)
                of, _ = TX.oracle_replay_grads(layers, B, L, seed, step, b, double=True, flip=[(gate, [near[0][0], near[0][1], c])])
                Dg, Co = gpu[name][c] - o64[name][c], of[name][c] - o64[name][c]
                frac = float((Dg * Co).sum() / (Co * Co).sum())
                resid = float((Dg - frac * Co).abs().max()) / scale
                print("      GPU row - oracle row (gate as in float64) = %.6f x (oracle row with the gate of token %s inverted - oracle row), residual %.2e"
                      % (frac, near[0], resid))
            if best:
                print("      best subset for this row: error %.3e with %s" % (best[0], [(g_, i_, float(pre[g_][tuple(i_)])) for g_, i_ in best[1]]))
                if best[0] < float(rowerr[c]) * 0.1:
                    flips += best[1]
    if flips:
        of, _ = TX.oracle_replay_grads(layers, B, L, seed, step, b, double=True, flip=flips)
        rows = worst(of)
        print("all %d flips together: worst %.3e at %s ; then %.3e %s" % ((len(flips),) + rows[0] + rows[1]))
        for g_, i_ in flips:
            print("   %s[%s]: float64 %.3e, fp32 oracle %.3e" % (g_, ",".join(map(str, i_)), float(pre[g_][tuple(i_)]), float(pre32[g_][tuple(i_)])))
    else:
        print("no flip set found")


if __name__ == "__main__":
    main()
