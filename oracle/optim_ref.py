"""TEST INFRASTRUCTURE -- transformers==3.0.2 optimizer + LR schedule restated.

The reference calls ``transformers.optimization.AdamW`` and
``get_linear_schedule_with_warmup`` (multimodal_driver.py:27-28, 345-350); their source
is a pinned third-party dependency (requirements.txt:348) that is NOT under
/root/reference and has no installed copy here (transformers 5.15 dropped AdamW).
"Parity unpinned by a library": this file restates the published 3.0.2 algorithm and
is pinned by the hand-derived known-answer vectors in tests/test_optim_oracle.py.

Published algorithm (3.0.2 AdamW.step, per parameter, per step t = 1, 2, ...):
    m <- b1*m + (1-b1)*g
    v <- b2*v + (1-b2)*g*g
    denom = sqrt(v) + eps                      (eps = 1e-6 default, added OUTSIDE the sqrt,
                                                BEFORE bias correction)
    step_size = lr * sqrt(1-b2^t) / (1-b1^t)   (correct_bias=True)
    p <- p - step_size * m / denom
    p <- p - lr*wd * p                         (decoupled decay, AFTER the update, uses the
                                                UPDATED p; skipped when wd == 0)
This is not torch.optim.AdamW (eps placement, decay order, eps default differ).
"""
import math

import torch


class AdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.0, correct_bias=True):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, correct_bias=correct_bias)
        super().__init__(params, defaults)

    @torch.no_grad()
    def step(self, closure=None):
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                g = p.grad
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p)
                    st["exp_avg_sq"] = torch.zeros_like(p)
                st["step"] += 1
                m, v = st["exp_avg"], st["exp_avg_sq"]
                m.mul_(b1).add_(g, alpha=1.0 - b1)
                v.mul_(b2).addcmul_(g, g, value=1.0 - b2)
                denom = v.sqrt().add_(group["eps"])
                step_size = group["lr"]
                if group["correct_bias"]:
                    step_size = step_size * math.sqrt(1.0 - b2 ** st["step"]) / (1.0 - b1 ** st["step"])
                p.addcdiv_(m, denom, value=-step_size)
                if group["weight_decay"] > 0.0:
                    p.add_(p, alpha=-group["lr"] * group["weight_decay"])


def linear_schedule_lambda(current_step, num_warmup_steps, num_training_steps):
    """3.0.2 get_linear_schedule_with_warmup's lr_lambda. num_warmup_steps may be a float
    (multimodal_driver.py:348: warmup_proportion * num_train_optimization_steps)."""
    if current_step < num_warmup_steps:
        return float(current_step) / float(max(1, num_warmup_steps))
    return max(0.0, float(num_training_steps - current_step) / float(max(1, num_training_steps - num_warmup_steps)))


def get_linear_schedule_with_warmup(optimizer, num_warmup_steps, num_training_steps, last_epoch=-1):
    return torch.optim.lr_scheduler.LambdaLR(
        optimizer, lambda s: linear_schedule_lambda(s, num_warmup_steps, num_training_steps), last_epoch)


def grouped_parameters(model, weight_decay=0.01):
    """multimodal_driver.py:328-343: substring match on bias / LayerNorm.bias / LayerNorm.weight."""
    no_decay = ["bias", "LayerNorm.bias", "LayerNorm.weight"]
    named = list(model.named_parameters())
    return [
        {"params": [p for n, p in named if not any(nd in n for nd in no_decay)], "weight_decay": weight_decay},
        {"params": [p for n, p in named if any(nd in n for nd in no_decay)], "weight_decay": 0.0},
    ]
