"""Single-node data parallelism for MAG-BERT fine-tuning: one process per GPU, RCCL over xGMI.

NEW relative to the reference, which is single-device (global_configs.py:4,7; DistributedSampler imported but unused,
multimodal_driver.py:21).  Every sample is independent (no cross-sample statistics; MSE is a batch mean), so the
minibatch shards across ranks and the only exchange is ONE logical all-reduce (sum) of the flat fp32 gradient
buffer per optimizer step.  It is issued in pieces on a side HIP stream while the backward is still running:

    backward stage s finishes (head | layer 11 | ... | layer 0 | MAG+embeddings)
        -> event on the compute stream -> comm stream waits -> all_reduce(flat_grads[range of stage s])

so layer-11 gradients travel while layer-10's dgrad/wgrad GEMMs run.  The flat layout makes each stage's weight
gradients one contiguous 28 MB range (no bucket copy-in/copy-out); the ~100 K small no-decay parameters (biases,
LayerNorm) plus the classifier go out as one final 0.4 MB piece.  The 1/world_size averaging is folded into the
AdamW kernel (AdamW.grad_scale), so no extra pass over the gradients.  xGMI is point-to-point (7 links/GPU): large
contiguous pieces let RCCL use all links; we never translate an NCCL bucket pattern.
"""
import os

import torch
import torch.distributed as dist


def shard_indices(n, rank, world, batch_size, seed, epoch, drop_last=False):
    """Per-rank minibatch index lists for one epoch.  A global shuffle (same on all ranks) is cut into global
    batches of world*batch_size; each rank takes its contiguous slice.  The ragged tail (n % (world*bs)) is split
    evenly so every rank runs the same number of steps (collectives stay matched); ranks may differ by one sample
    in the last step, which the SUM/world averaging weights by 1/world per rank (stated in DESIGN.md)."""
    g = torch.Generator()
    g.manual_seed(int(seed) * 1000003 + int(epoch))
    perm = torch.randperm(n, generator=g).tolist()
    gb = world * batch_size
    out = []
    full = n // gb
    for i in range(full):
        base = i * gb + rank * batch_size
        out.append(perm[base: base + batch_size])
    tail = perm[full * gb:]
    if tail and not drop_last:
        per = (len(tail) + world - 1) // world
        mine = tail[rank * per: (rank + 1) * per]
        if len(tail) >= world:          # every rank gets >= 1 sample
            out.append(mine)
    return out


class GradReducer(object):
    """All-reduce(sum) of ranges of a flat gradient tensor on a side stream (CUDA) or inline (CPU/gloo)."""

    def __init__(self, flat_grads, process_group=None, wire_dtype=torch.float32):
        """wire_dtype: torch.float32 (exact: the sum of the ranks' fp32 gradients) or torch.bfloat16 (bf16 perf mode: halves the
        bytes on the xGMI links -- with 2 GPUs the 443 MB fp32 exchange over one link takes as long as the whole backward;
        gradients are rounded to bf16 before the sum, moments and parameters stay fp32)."""
        self.g = flat_grads
        self.wire_dtype = wire_dtype
        self.stage_buf = None
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # MB_DP_FORCE=1: issue the collectives even in a 1-rank group (exercises the RCCL call path on a single GPU; tests)
        self.active = self.world > 1 or (dist.is_initialized() and os.environ.get("MB_DP_FORCE") == "1")
        self.cuda = flat_grads.is_cuda
        self.comm_stream = torch.cuda.Stream(device=flat_grads.device) if self.cuda else None
        self.pending = []

    def reduce_ranges(self, ranges):
        if not self.active or not ranges:
            return
        if self.cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.g.device))
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(ev)
                if self.wire_dtype == torch.bfloat16:
                    from . import _lib
                    L = _lib.lib()
                    if self.stage_buf is None:
                        self.stage_buf = torch.empty(self.g.numel(), dtype=torch.bfloat16, device=self.g.device)
                    cs = self.comm_stream.cuda_stream
                    for off, n in ranges:
                        _lib.check(L.mb_narrow(_lib.DT_BF16, self.g.data_ptr() + 4 * off, self.stage_buf.data_ptr() + 2 * off, n, cs))
                        dist.all_reduce(self.stage_buf[off: off + n], op=dist.ReduceOp.SUM, group=self.pg)
                        _lib.check(L.mb_widen(_lib.DT_BF16, self.stage_buf.data_ptr() + 2 * off, self.g.data_ptr() + 4 * off, n, cs))
                else:
                    for off, n in ranges:
                        dist.all_reduce(self.g[off: off + n], op=dist.ReduceOp.SUM, group=self.pg)
        else:
            for off, n in ranges:
                dist.all_reduce(self.g[off: off + n], op=dist.ReduceOp.SUM, group=self.pg)

    def mark(self):
        """an event behind every piece enqueued so far (None on the host path, where reduce_ranges is synchronous)"""
        if not (self.cuda and self.active):
            return None
        ev = torch.cuda.Event()
        ev.record(self.comm_stream)
        return ev

    def wait(self):
        """make the compute stream wait for every outstanding piece (call before optimizer.step())"""
        if self.cuda and self.active:
            torch.cuda.current_stream(self.g.device).wait_stream(self.comm_stream)


def complement(ranges, n):
    """sorted disjoint (off, len) pieces of [0, n) that no range in `ranges` covers"""
    out, cur = [], 0
    for off, ln in sorted(ranges):
        if off > cur:
            out.append((cur, off - cur))
        cur = max(cur, off + ln)
    if cur < n:
        out.append((cur, n - cur))
    return out


def stage_plan(core):
    """([large ranges of stage s] for every backward stage, [the rest]) from the engine's layout.  The rest -- small tensors
    that no stage reports as a large range: biases, LayerNorms, the classifier -- is whatever the large ranges leave uncovered,
    so every trainable element is reduced exactly once whatever the engine reports (MAG-XLNet hands over its whole no-decay
    block as one large range of its last stage; MAG-BERT's per-layer no-decay spans are small and end up here)."""
    nstage = core.n_layers + 2
    plan = []
    for s in range(nstage):
        plan.append([(off, n) for off, n in core.stage_ranges(s) if n >= (1 << 16)])
    seen = [r for big in plan for r in big]
    for i, (o1, n1) in enumerate(sorted(seen)):
        for o2, n2 in sorted(seen)[i + 1:]:
            if o2 < o1 + n1:
                raise RuntimeError("engine stage ranges overlap: (%d,%d) and (%d,%d)" % (o1, n1, o2, n2))
    return plan, complement(seen, core.n_params)


class DataParallel(object):
    """Wraps a MAG_BertForSequenceClassification: hooks the engine's backward stages to the reducer."""

    def __init__(self, model, optimizer=None, process_group=None):
        self.model = model
        self.core = model._core
        # Wire format of the gradient exchange.  auto: bf16 only where the links are the bound -- bf16 perf mode, RCCL, and a
        # 2-GPU group (xGMI is point-to-point: two GPUs share ONE ~150 GB/s link, so the 443 MB fp32 exchange lasts as long as
        # the whole backward; with 4 / 8 GPUs a ring spreads over 3 / 7 links and the fp32 exchange hides under the backward,
        # while the two conversion passes would cost ~4 % of the step -- measured with a 1-rank RCCL group).
        wire = os.environ.get("MB_DP_GRAD_DTYPE", "auto")
        if wire == "auto":
            on_rccl = dist.is_initialized() and dist.get_backend(process_group) == "nccl"
            few_links = dist.is_initialized() and dist.get_world_size(process_group) <= 2
            wire = "bf16" if (self.core.compute_dtype == torch.bfloat16 and self.core.grads.is_cuda and on_rccl and few_links) else "fp32"
        self.reducer = GradReducer(self.core.grads, process_group, torch.bfloat16 if wire == "bf16" else torch.float32)
        self.world = self.reducer.world
        self.plan, self.tail = stage_plan(self.core)
        self.core.stage_hooks.insert(0, self._on_stage)
        self.sync = True          # set False on gradient-accumulation micro-steps (multimodal_driver.py:383)
        self.optimizer = optimizer
        self.late_ranges = []
        self.split_last = os.environ.get("MB_DP_SPLIT_LAST", "1") != "0"
        # exposed communication: timing events around the two places where the compute stream waits for the comm stream
        self._tev = [torch.cuda.Event(enable_timing=True) for _ in range(4)] if self.core.grads.is_cuda else None
        self._tev_used = [False, False]
        if optimizer is not None:
            optimizer.grad_scale = 1.0 / self.world
            optimizer._dp = self

    def broadcast_parameters(self, src=0):
        if self.reducer.active:
            dist.broadcast(self.core.params, src=src, group=self.reducer.pg)
            self.core.weights_dirty = True

    def ready_ranges(self, stage):
        """flat gradient ranges whose all-reduce has been enqueued once `stage` is done"""
        r = list(self.plan[stage])
        if stage == len(self.plan) - 1:
            r.extend(self.tail)
        return r

    def _on_stage(self, stage):
        if not self.sync:
            return
        if stage < len(self.plan) - 1:
            self.reducer.reduce_ranges(self.plan[stage])
            return
        # last stage: its pieces (embeddings + MAG: 94 MB, and the small tail) are produced last and would be fully exposed.
        # The compute stream only waits for everything BEFORE them; AdamW.step() updates the already-reduced ranges under this
        # last all-reduce and calls finish() before it touches the late ranges (split_last = False: plain full wait here).
        early = self.reducer.mark()
        self.reducer.reduce_ranges(self.plan[stage])
        self.reducer.reduce_ranges(self.tail)
        if self.split_last and early is not None and self.optimizer is not None:
            self._timed_wait(0, lambda cs: cs.wait_event(early))
            self.late_ranges = [r for r in list(self.plan[stage]) + list(self.tail) if r[1] > 0]
        else:
            self._timed_wait(0, lambda cs: self.reducer.wait())
            self._tev_used[1] = False

    def _timed_wait(self, k, wait):
        """run `wait` (which makes the compute stream wait for gradient pieces) between two timing events on that stream"""
        cs = torch.cuda.current_stream(self.core.grads.device) if self.core.grads.is_cuda else None
        if self._tev is None or not self.reducer.active:
            wait(cs)
            return
        self._tev[2 * k].record(cs)
        wait(cs)
        self._tev[2 * k + 1].record(cs)
        self._tev_used[k] = True

    def exposed_ms(self):
        """Time the compute stream of the LAST step spent stalled on the gradient exchange (both waits: before the early
        AdamW ranges and before the late ones); synchronises.  0 when every piece had landed by the time it was needed."""
        if self._tev is None:
            return 0.0
        total = 0.0
        for k in range(2):
            if self._tev_used[k]:
                self._tev[2 * k + 1].synchronize()
                total += self._tev[2 * k].elapsed_time(self._tev[2 * k + 1])
        return total

    def finish(self):
        """full wait for the gradient exchange (AdamW.step() calls it before the late ranges; harmless to call twice)"""
        if self.late_ranges:
            self._timed_wait(1, lambda cs: self.reducer.wait())
        else:
            self.reducer.wait()
        self.late_ranges = []

    def __getattr__(self, name):
        return getattr(self.model, name)

    def __call__(self, *a, **k):
        return self.model(*a, **k)
