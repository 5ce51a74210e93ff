"""CPU: hand-derived known answers for transformers==3.0.2 AdamW + linear warmup schedule (oracle/optim_ref.py and
the product's host-side schedule).  No installed library holds this formula, so the pin is
arithmetic written out from the published algorithm."""
import math

import torch

from oracle import optim_ref as O
from bert_multimodal_transformer_amd.optimization import linear_schedule_lambda


def _hand(p, g_seq, lr_seq, wd, b1=0.9, b2=0.999, eps=1e-6):
    m = v = 0.0
    for t, (g, lr) in enumerate(zip(g_seq, lr_seq), start=1):
        m = b1 * m + (1 - b1) * g
        v = b2 * v + (1 - b2) * g * g
        step = lr * math.sqrt(1 - b2 ** t) / (1 - b1 ** t)
        p = p - step * m / (math.sqrt(v) + eps)
        if wd > 0:
            p = p - lr * wd * p
    return p


def test_adamw_three_steps_two_groups():
    w = torch.nn.Parameter(torch.tensor([0.5, -1.0]))
    b = torch.nn.Parameter(torch.tensor([0.25]))
    opt = O.AdamW([{"params": [w], "weight_decay": 0.01}, {"params": [b], "weight_decay": 0.0}], lr=1e-2)
    sched = O.get_linear_schedule_with_warmup(opt, num_warmup_steps=0.1 * 20, num_training_steps=20)
    gw = [[0.1, -0.2], [0.3, 0.05], [-0.4, 0.6]]
    gb = [[1.0], [-2.0], [0.5]]
    lrs = []
    for t in range(3):
        w.grad = torch.tensor(gw[t]); b.grad = torch.tensor(gb[t])
        lrs.append(opt.param_groups[0]["lr"])
        opt.step(); sched.step()
    # schedule: warmup = 2.0 steps -> lambda(0)=0, lambda(1)=0.5, lambda(2)=(20-2)/(20-2)=1
    assert lrs == [0.0, 0.005, 0.01]
    for i in range(2):
        assert abs(float(w[i]) - _hand([0.5, -1.0][i], [g[i] for g in gw], lrs, 0.01)) < 1e-6
    assert abs(float(b[0]) - _hand(0.25, [g[0] for g in gb], lrs, 0.0)) < 1e-6
    # first real update (t=2, lr=0.005): p - lr*sign-ish.  closed form for one element, written out:
    m1 = 0.1 * 0.1; v1 = 0.001 * 0.01
    m2 = 0.9 * m1 + 0.1 * 0.3; v2 = 0.999 * v1 + 0.001 * 0.09
    p2 = 0.5 - 0.005 * math.sqrt(1 - 0.999 ** 2) / (1 - 0.9 ** 2) * m2 / (math.sqrt(v2) + 1e-6)
    p2 = p2 - 0.005 * 0.01 * p2
    assert abs(_hand(0.5, [0.1, 0.3], [0.0, 0.005], 0.01) - p2) < 1e-12


def test_linear_schedule_matches_driver_arithmetic():
    # MOSI: int(1281/48/1)*40 = 1040 optimisation steps, warmup 0.1*1040 = 104.0 (float), loop runs 27*40 = 1080
    T = int(1281 / 48 / 1) * 40
    assert T == 1040
    w = 0.1 * T
    for f in (O.linear_schedule_lambda, linear_schedule_lambda):
        assert f(0, w, T) == 0.0
        assert abs(f(52, w, T) - 0.5) < 1e-12
        assert f(104, w, T) == 1.0
        assert abs(f(572, w, T) - 0.5) < 1e-12
        assert f(1040, w, T) == 0.0 and f(1079, w, T) == 0.0      # lr stays 0 for the last 40 steps
