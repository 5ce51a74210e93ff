"""AdamW + get_linear_schedule_with_warmup -- the transformers==3.0.2 API the reference imports
(/root/reference/multimodal_driver.py:27-28, 345-350), running on the fused HIP kernel (csrc/adamw.hip).

    optimizer = AdamW(optimizer_grouped_parameters, lr=args.learning_rate)
    scheduler = get_linear_schedule_with_warmup(optimizer, num_warmup_steps=..., num_training_steps=...)

Parameters that are views of a model's flat buffer (bert.py) are updated with ONE launch per param group over the
contiguous flat range (m, v live in flat buffers too; the bf16 operand shadow is refreshed and the gradient cleared
in the same pass); any other CUDA tensor gets one launch per tensor.  Formula: see oracle/optim_ref.py -- eps 1e-6
outside the sqrt, bias correction folded into step_size, decoupled decay AFTER the update.
"""
import torch

from . import _lib


class AdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.0, correct_bias=True):
        if lr < 0.0:
            raise ValueError("Invalid learning rate: {} - should be >= 0.0".format(lr))
        if not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0:
            raise ValueError("Invalid beta parameters: {}".format(betas))
        if not 0.0 <= eps:
            raise ValueError("Invalid epsilon value: {} - should be >= 0.0".format(eps))
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, correct_bias=correct_bias)
        super().__init__(params, defaults)
        self.grad_scale = 1.0          # data parallel: 1/world_size when gradients were SUM-reduced
        self.fused_zero_grad = True    # clear gradients inside the update kernel (zero_grad() then costs nothing)
        self._plan = None
        self._t = 0
        self._dp = None              # set by distributed.DataParallel

    # -- planning: map groups onto contiguous flat ranges ------------------------------------------------
    def _build_plan(self):
        plan = []
        for gi, group in enumerate(self.param_groups):
            flat, loose = {}, []
            for p in group["params"]:
                info = getattr(p, "_mb_flat", None)
                if info is None:
                    loose.append(p)
                else:
                    core, off, numel, _ = info
                    flat.setdefault(id(core), (core, []))[1].append((off, numel))
            for core, spans in flat.values():
                spans.sort()
                merged = []
                for off, numel in spans:
                    end = (off + numel + 63) // 64 * 64          # tensors are 64-float aligned in the flat layout
                    if merged and off <= merged[-1][1]:
                        merged[-1][1] = max(merged[-1][1], end)
                    else:
                        merged.append([off, end])
                if not hasattr(core, "_adam_m"):
                    core._adam_m = torch.zeros_like(core.params)
                    core._adam_v = torch.zeros_like(core.params)
                for a, b in merged:
                    plan.append(("flat", gi, core, a, min(b, core.n_params)))
            for p in loose:
                plan.append(("loose", gi, p))
        self._plan = plan
        # cores whose GEMM-weight range [sh_begin, sh_end) is updated -- hence zeroed -- entirely by this optimizer: after a
        # fused step their layer weight gradients are known to be zero (mb_bert_mark_grads_zero: the next backward stores them)
        self._covers = {}
        spans = {}
        for it in plan:
            if it[0] == "flat":
                spans.setdefault(id(it[2]), (it[2], []))[1].append((it[3], it[4]))
        for cid, (core, sp) in spans.items():
            sp.sort()
            reach = core.sh_begin
            for a, b in sp:
                if a <= reach:
                    reach = max(reach, b)
            self._covers[cid] = (core, reach >= core.sh_end)

    def _mark_zero(self):
        for core, covered in self._covers.values():
            if covered:
                core.mark_grads_zero(True)

    def flat_step_args(self, core, allow_dp=False):
        """Hyper-parameters of step() as ONE whole-buffer update, when this optimizer is exactly the driver's two parameter
        groups (multimodal_driver.py:329-343) over `core`'s flat buffer: [0, n_decay) decayed, the rest not, same lr / betas /
        eps / bias correction in both.  None otherwise (loose tensors, more groups, diverged groups, data parallel): the caller
        then runs step() as usual.  Used by the whole-step graph (mb_bert_train_step), which applies the update itself."""
        if (self._dp is not None and not allow_dp) or not self.fused_zero_grad:
            return None
        if self._plan is None:
            self._build_plan()
        flats = [it for it in self._plan if it[0] == "flat"]
        loose = [it for it in self._plan if it[0] != "flat"]
        if any(it[2].grad is not None for it in loose):           # grad-less parameters (MAG-XLNet's frozen mask_emb) are skipped by step() too
            return None
        if len(flats) != 2 or any(it[2] is not core for it in flats):
            return None
        (_, g0, _, a0, b0), (_, g1, _, a1, b1) = sorted(flats, key=lambda it: it[3])
        if (a0, b0, a1, b1) != (0, core.n_decay, core.n_decay, getattr(core, "n_update_end", core.n_params)):
            return None
        ga, gb = self.param_groups[g0], self.param_groups[g1]
        if any(ga[k] != gb[k] for k in ("lr", "betas", "eps", "correct_bias")) or gb["weight_decay"] != 0.0:
            return None
        return dict(m=core._adam_m, v=core._adam_v, lr=float(ga["lr"]), beta1=float(ga["betas"][0]), beta2=float(ga["betas"][1]),
                    eps=float(ga["eps"]), weight_decay=float(ga["weight_decay"]), correct_bias=bool(ga["correct_bias"]),
                    grad_scale=float(self.grad_scale))

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        if self._plan is None:
            self._build_plan()
        L = _lib.lib()
        self._t += 1
        for core, _ in self._covers.values():        # gradients a fused step left "logically zero" become real zeros before they are read
            core.materialize_grads()
        dp = self._dp
        late = []
        if dp is not None and getattr(dp, "late_ranges", None):
            # data parallel: the all-reduce of the LAST backward stage (word embeddings: 94 MB, produced last) is still on the
            # wire.  Everything else is already reduced (the stage hook made this stream wait for the "early" marker only), so
            # the update of those ranges (~78 % of the parameters) runs under that last all-reduce; the late ranges follow after
            # the full wait.
            late = sorted(dp.late_ranges)

        shards = getattr(dp, "shards", None) if dp is not None else None

        def launch(core, group, x, y):
            if shards is not None and core is shards.core and x < shards.hi and y > shards.lo:
                # sharded update (distributed.OptimizerShards): of the layers' GEMM weights only this rank's shard; the rest of
                # [x, y) -- in front of / behind the sharded range -- is replicated as usual
                pieces = [(x, min(y, shards.lo)), (max(x, shards.a), min(y, shards.b)), (max(x, shards.hi), y)]
                for px, py in pieces:
                    if py > px:
                        launch_range(core, group, px, py)
                return
            launch_range(core, group, x, y)

        def launch_range(core, group, x, y):
            if y <= x:
                return
            b1, b2 = group["betas"]
            sh = core.shadow if core.dt == _lib.DT_BF16 else None
            sb = min(max(core.sh_begin, x), y) - x
            se = min(max(core.sh_end, x), y) - x
            _lib.check(L.mb_adamw_step(
                core.params.data_ptr() + 4 * x, core.grads.data_ptr() + 4 * x, core._adam_m.data_ptr() + 4 * x,
                core._adam_v.data_ptr() + 4 * x, (sh.data_ptr() + 2 * x) if sh is not None else None, y - x,
                (y - x) if group["weight_decay"] > 0.0 else 0, sb, se, group["lr"], b1, b2, group["eps"],
                group["weight_decay"], self._t, 1 if group["correct_bias"] else 0, self.grad_scale,
                1 if self.fused_zero_grad else 0, core.stream()))

        deferred = []                      # (core, group, x, y) pieces that must wait for the last all-reduce
        for item in self._plan:
            group = self.param_groups[item[1]]
            b1, b2 = group["betas"]
            if item[0] == "flat":
                _, _, core, a, b = item
                cur = a
                for lo, n in late:         # split [a, b) around the late ranges (sorted, disjoint, tensor-aligned)
                    hi = lo + n
                    if hi <= cur or lo >= b or core is not dp.core:
                        continue
                    launch(core, group, cur, max(cur, lo))
                    deferred.append((core, group, max(cur, lo), min(b, hi)))
                    cur = min(b, hi)
                launch(core, group, cur, b)
            else:
                p = item[2]
                if p.grad is None:
                    continue
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                    raise _lib.MagbertError("AdamW runs on the HIP kernel only: fp32 contiguous ROCm tensors required")
                state = self.state[p]
                if len(state) == 0:
                    state["exp_avg"] = torch.zeros_like(p)
                    state["exp_avg_sq"] = torch.zeros_like(p)
                g = p.grad.contiguous()
                n = p.numel()
                _lib.check(L.mb_adamw_step(p.data_ptr(), g.data_ptr(), state["exp_avg"].data_ptr(),
                                           state["exp_avg_sq"].data_ptr(), None, n, n if group["weight_decay"] > 0.0 else 0,
                                           0, 0, group["lr"], b1, b2, group["eps"], group["weight_decay"], self._t,
                                           1 if group["correct_bias"] else 0, self.grad_scale, 0,
                                           torch.cuda.current_stream(p.device).cuda_stream))
        if late:
            dp.finish()                    # this stream now waits for the last all-reduce
            for core, group, x, y in deferred:
                launch(core, group, x, y)
        if shards is not None:
            shards.clear_dead_gradients()  # (before the flag below: the buffer really is all zeros again)
            shards.gather_updated()
        if self.fused_zero_grad:
            self._mark_zero()              # every gradient the update consumed is zero again
        return loss

    # -- checkpoint / resume ---------------------------------------------------------
    def state_dict(self):
        """torch.optim.Optimizer.state_dict() plus what lives outside `self.state`: the step count and, per model, the flat
        Adam moment buffers (tensors on the host, keyed by position among the distinct flat buffers of the plan)."""
        if self._plan is None:
            self._build_plan()
        sd = super().state_dict()
        cores = []
        for item in self._plan:
            if item[0] == "flat" and all(item[2] is not c for c in cores):
                cores.append(item[2])
        # sharded optimizer update under data parallel (MB_DP_SHARD_OPT=1): a rank's moments outside its own slices are stale until
        # gathered.  COLLECTIVE in that mode -- every rank must call state_dict() ("Checkpoints under data parallel").
        for c in cores:
            c.refresh_sharded_state(adam=True)
        sd["magbert"] = {"t": self._t,
                         "flat": [{"exp_avg": c._adam_m.detach().cpu(), "exp_avg_sq": c._adam_v.detach().cpu()} for c in cores]}
        return sd

    def load_state_dict(self, state_dict):
        state_dict = dict(state_dict)
        extra = state_dict.pop("magbert", None)
        super().load_state_dict(state_dict)
        self._plan = None
        self._build_plan()
        if extra is not None:
            self._t = int(extra["t"])
            cores = []
            for item in self._plan:
                if item[0] == "flat" and all(item[2] is not c for c in cores):
                    cores.append(item[2])
            if len(cores) != len(extra["flat"]):
                raise ValueError("optimizer checkpoint holds %d flat buffers, this optimizer has %d" % (len(extra["flat"]), len(cores)))
            for c, st in zip(cores, extra["flat"]):
                c._adam_m.copy_(st["exp_avg"])
                c._adam_v.copy_(st["exp_avg_sq"])

    def zero_grad(self, set_to_none=False):
        if self._plan is None:
            self._build_plan()
        done = set()
        for item in self._plan:
            if item[0] == "flat":
                core = item[2]
                if not self.fused_zero_grad and id(core) not in done:
                    core.grads.zero_()
                    core.mark_grads_zero(True)
                    done.add(id(core))
            else:
                p = item[2]
                if p.grad is not None:
                    p.grad.zero_()


def linear_schedule_lambda(current_step, num_warmup_steps, num_training_steps):
    """lr multiplier of transformers 3.0.2 get_linear_schedule_with_warmup (num_warmup_steps may be a float:
    multimodal_driver.py:348 passes warmup_proportion * num_train_optimization_steps)."""
    if current_step < num_warmup_steps:
        return float(current_step) / float(max(1, num_warmup_steps))
    return max(0.0, float(num_training_steps - current_step) / float(max(1, num_training_steps - num_warmup_steps)))


def get_linear_schedule_with_warmup(optimizer, num_warmup_steps, num_training_steps, last_epoch=-1):
    return torch.optim.lr_scheduler.LambdaLR(
        optimizer, lambda s: linear_schedule_lambda(s, num_warmup_steps, num_training_steps), last_epoch)
