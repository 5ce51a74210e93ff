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


def tail_sizes(n_tail, world):
    """how many samples of a ragged tail of n_tail samples each rank takes: as even as possible, the first n_tail % world ranks one
    more (every rank gets >= 1 whenever n_tail >= world)"""
    base, rem = divmod(n_tail, world)
    return [base + (1 if r < rem else 0) for r in range(world)]


def shard_indices(n, rank, world, batch_size, seed, epoch, drop_last=False):
    """Per-rank minibatch index lists for one epoch.  A global shuffle (same on all ranks) is cut into global
    batches of world*batch_size; each rank takes its contiguous slice.  The ragged tail (n % (world*bs)) is split
    as evenly as possible so every rank runs the same number of steps (collectives stay matched); ranks may differ by one
    sample in the last step -- shard_step_sizes() says by how much, and the driver weights each rank's loss by its share
    (loss_scale = B_rank * world / B_global), so the averaged gradient is the mean over the global batch."""
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
    if tail and not drop_last and len(tail) >= world:          # every rank gets >= 1 sample (a shorter tail is dropped)
        sizes = tail_sizes(len(tail), world)
        lo = sum(sizes[:rank])
        out.append(tail[lo: lo + sizes[rank]])
    return out


def shard_step_sizes(n, world, batch_size, drop_last=False):
    """[step] -> per-rank batch sizes of shard_indices' plan (the same on every rank, no communication)"""
    gb = world * batch_size
    out = [[batch_size] * world for _ in range(n // gb)]
    t = n % gb
    if t and not drop_last and t >= world:
        out.append(tail_sizes(t, world))
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
        # fork / mark events are created once and re-recorded (32 > pieces per step): no per-step event churn
        self._events = [torch.cuda.Event() for _ in range(32)] if self.cuda else []
        self._next_event = 0

    def _event(self):
        ev = self._events[self._next_event]
        self._next_event = (self._next_event + 1) % len(self._events)
        return ev

    def reduce_ranges(self, ranges):
        if not self.active or not ranges:
            return
        if self.cuda:
            ev = self._event()
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
        ev = self._event()
        ev.record(self.comm_stream)
        return ev

    def wait(self):
        """make the compute stream wait for every outstanding piece (call before optimizer.step())"""
        if self.cuda and self.active:
            torch.cuda.current_stream(self.g.device).wait_stream(self.comm_stream)


def exchange_embedding_rows(grad_table, ids, capacity, group=None):
    """Sum over the ranks of a word-embedding gradient [V, H] of which every rank touched at most `capacity` rows (the token
    ids of its minibatch), moving only those rows: 2 all-gathers of capacity x (8 + 4H) bytes instead of a dense all-reduce
    of V x H x 4 (bert-base: 7.4 MB instead of 93.8 MB at 2,400 tokens per rank).  In place; the result is the dense sum.

    Every rank sends its sorted ids and the matching gradient rows (a repeated id sends its row once, the repeats send zeros;
    the tail up to `capacity` repeats the last id with zeros), then clears every row any rank touched and adds the
    contributions in rank order -- the same additions in the same order everywhere, so the replicas stay bit-identical."""
    world = dist.get_world_size(group)
    flat = ids.reshape(-1).to(grad_table.device)
    n = flat.numel()
    if n > capacity:
        raise RuntimeError("exchange_embedding_rows: %d ids exceed the agreed capacity %d" % (n, capacity))
    srt, _ = torch.sort(flat)
    if n < capacity:
        srt = torch.cat([srt, srt[-1:].expand(capacity - n)])
    first = torch.ones(capacity, dtype=torch.bool, device=srt.device)
    first[1:] = srt[1:] != srt[:-1]
    vals = grad_table.index_select(0, srt) * first.unsqueeze(1).to(grad_table.dtype)
    rows_all = [torch.empty_like(srt) for _ in range(world)]
    vals_all = [torch.empty_like(vals) for _ in range(world)]
    dist.all_gather(rows_all, srt, group=group)
    dist.all_gather(vals_all, vals, group=group)
    grad_table.index_fill_(0, torch.cat(rows_all), 0.0)
    for r in range(world):
        grad_table.index_add_(0, rows_all[r], vals_all[r])
    return grad_table


class Comm(object):
    """The gradient exchange of the single-call data-parallel step (include/magbert_hip.h: mb_comm, csrc/comm.hip): a C object that
    owns the comm stream and issues the collectives itself.  Backend "nccl" -> RCCL called from C (rank 0's ncclUniqueId travels
    over torch.distributed once); any other backend (gloo: the two-ranks-on-one-GPU tests) -> host callbacks that run
    torch.distributed collectives on the comm stream."""

    def __init__(self, core, process_group=None, wire_dtype=torch.float32, sparse_rows=True, row_capacity=0):
        import ctypes as C
        from . import _lib
        self._lib = _lib
        L = _lib.lib()
        self.core, self.pg = core, process_group
        self.rank, self.world = dist.get_rank(process_group), dist.get_world_size(process_group)
        self.backend = dist.get_backend(process_group)
        dev = core.device
        h = C.c_void_p()
        if self.backend == "nccl":
            idbuf = (C.c_uint8 * 128)()
            # on EVERY rank (only rank 0's id is used): RCCL loads -- or fails to -- everywhere before the first collective below.  A rank
            # whose library is missing must not raise yet (the others would sit in the broadcast forever): the return codes
            # are agreed over the group first, then every rank falls back together
            rc0 = L.mb_comm_unique_id(idbuf)
            ok = torch.tensor([1 if rc0 == 0 else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=process_group)
            if int(ok.item()) == 0:
                if rc0 != 0:
                    _lib.check(rc0)
                raise RuntimeError("another rank could not load RCCL")
            t = torch.tensor(list(idbuf), dtype=torch.uint8, device=dev)
            dist.broadcast(t, src=dist.get_global_rank(process_group, 0) if process_group is not None else 0, group=process_group)
            raw = bytes(t.cpu().tolist())
            rc = L.mb_comm_create_rccl(C.c_char_p(raw), self.rank, self.world, C.byref(h))
            ok = torch.tensor([1 if rc == 0 else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=process_group)      # all ranks use the C-side exchange, or none does
            if int(ok.item()) == 0:
                if rc == 0:
                    L.mb_comm_destroy(h)
                    raise RuntimeError("another rank could not create its RCCL communicator")
                _lib.check(rc)
            self._cbs = None
        else:
            self._cbs = (_lib.ALL_REDUCE_CB(self._all_reduce_cb), _lib.ALL_GATHER_CB(self._all_gather_cb))      # kept alive with the object
            _lib.check(L.mb_comm_create_callbacks(self.rank, self.world, C.cast(self._cbs[0], C.c_void_p), C.cast(self._cbs[1], C.c_void_p),
                                                  None, C.byref(h)))
        self.handle = h
        self.wire_dtype = wire_dtype
        wire = _lib.DT_BF16 if wire_dtype == torch.bfloat16 else _lib.DT_F32
        vocab = H = 0
        if sparse_rows:
            for name, off, numel, shape, decay in core.tensors:
                if name.endswith("word_embeddings.weight") or name.endswith("word_embedding.weight"):
                    vocab, H = int(shape[0]), int(shape[1])
        self.sparse = vocab > 0
        # per-rank row capacity of the word-embedding exchange: agreed once over the group (the largest engine of the group)
        cap = torch.tensor([max(core.max_B * core.max_L, int(row_capacity), int(os.environ.get("MB_DP_ROW_CAPACITY", "0")))], dtype=torch.int64,
                           device=dev if self.backend == "nccl" else "cpu")
        dist.all_reduce(cap, op=dist.ReduceOp.MAX, group=process_group)
        self.capacity = int(cap.item()) if self.sparse else 0
        n = int(core.n_params)
        nbytes = L.mb_comm_scratch_bytes(self.world, wire, n, vocab, H, self.capacity)
        self.scratch = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        _lib.check(L.mb_comm_bind_scratch(h, _lib.ptr(self.scratch), nbytes, wire, n, vocab, H, self.capacity))
        self.sharding = False
        self.stream = torch.cuda.ExternalStream(L.mb_comm_stream(h), device=dev)
        self.stream.synchronize()          # (the slot table's one-time memset)
        torch.cuda.synchronize(dev)        # the set-up collectives above are complete before the step that follows starts capturing

    # -- callback backend -------------------------------------------------------------------------------------------------------
    def _view(self, ptr, nbytes):
        flat = [self.core.grads, self.scratch, self.core.params, getattr(self.core, "shadow", None), getattr(self.core, "_adam_m", None),
                getattr(self.core, "_adam_v", None)]          # (the sharded update gathers parameters / the bf16 shadow / Adam moments)
        for t in flat:
            if t is None or t.numel() < 2:
                continue
            base = t.data_ptr()
            if base <= ptr and ptr + nbytes <= base + t.numel() * t.element_size():
                return t.view(torch.uint8)[ptr - base: ptr - base + nbytes] if t.dtype == torch.uint8 else \
                    t.view(-1).view(torch.uint8)[ptr - base: ptr - base + nbytes]
        raise RuntimeError("collective over memory that is none of the engine's flat buffers nor the comm scratch")

    def _stream_of(self, stream):
        """torch stream object of a raw hipStream_t handed to a callback (NULL = the device's default stream)"""
        return torch.cuda.ExternalStream(stream, device=self.core.device) if stream else torch.cuda.default_stream(self.core.device)

    def _all_reduce_cb(self, ctx, buf, count, dtype, stream):
        try:
            tdt = torch.bfloat16 if dtype == self._lib.DT_BF16 else torch.float32
            v = self._view(buf, count * (2 if tdt == torch.bfloat16 else 4)).view(tdt)
            with torch.cuda.stream(self._stream_of(stream)):
                dist.all_reduce(v, op=dist.ReduceOp.SUM, group=self.pg)
            return 0
        except Exception:          # an exception must not unwind through the C frames
            import traceback
            traceback.print_exc()
            return 1005

    def _all_gather_cb(self, ctx, buf, bytes_per_rank, stream):
        try:
            v = self._view(buf, bytes_per_rank * self.world)
            parts = [v[r * bytes_per_rank: (r + 1) * bytes_per_rank] for r in range(self.world)]
            with torch.cuda.stream(self._stream_of(stream)):
                dist.all_gather(parts, parts[self.rank].clone(), group=self.pg)
            return 0
        except Exception:
            import traceback
            traceback.print_exc()
            return 1005

    # -- modes ------------------------------------------------------------------------------------------------------------------
    def set_row_exchange(self, rowwise):
        """False for the step that ends a gradient-accumulation window (the table's rows are the union over the micro-steps)"""
        self._lib.check(self._lib.lib().mb_comm_set_row_exchange(self.handle, 1 if rowwise else 0))

    def set_sharding(self, on):
        """sharded optimizer update inside the single-call step (include/magbert_hip.h: mb_comm_set_sharding)"""
        self._lib.check(self._lib.lib().mb_comm_set_sharding(self.handle, 1 if on else 0))
        self.sharding = bool(self._lib.lib().mb_comm_sharding(self.handle))

    def shard_slices(self):
        """this rank's [begin, end) slices of the flat buffers in the last sharded step"""
        import ctypes as C
        buf = (C.c_size_t * 64)()
        n = self._lib.lib().mb_comm_shard_slices(self.handle, buf, 32)
        return [(buf[2 * i], buf[2 * i + 1]) for i in range(n)]

    # -- measurement ------------------------------------------------------------------------------------------------------------
    def set_timing(self, on):
        self._lib.check(self._lib.lib().mb_comm_set_timing(self.handle, 1 if on else 0))

    def exposed_ms(self):
        import ctypes as C
        v = C.c_float()
        self._lib.check(self._lib.lib().mb_comm_exposed_ms(self.handle, C.byref(v)))
        return float(v.value)

    def stats(self):
        """(collectives issued, bytes handed to them) in the last step"""
        import ctypes as C
        a, b = C.c_size_t(), C.c_size_t()
        self._lib.check(self._lib.lib().mb_comm_stats(self.handle, C.byref(a), C.byref(b)))
        return a.value, b.value

    def close(self):
        if getattr(self, "handle", None) is not None:
            self._lib.lib().mb_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


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


class OptimizerShards(object):
    """Sharded optimizer update under data parallel (ZeRO-1 for the part of the model where it pays).  NEW relative to the
    reference (single process: /root/reference/multimodal_driver.py:345, 384-386 run one AdamW over everything).

    With N replicas the plain scheme runs N identical AdamW sweeps: 30 B/parameter of HBM traffic per rank and step, 16 % of the
    step, at the kernel's HBM roof.  Here the layers' GEMM weights -- [sh_begin, sh_end), 85.5 M of the 110.9 M parameters, the
    range that has a bf16 operand shadow -- are cut into N contiguous shards: the gradient pieces of that range are REDUCED TO
    THEIR OWNER (reduce-scatter: half the wire bytes of an all-reduce), the owner alone updates its shard (p, m, v of the other
    shards are not touched: -0.45 ms of HBM traffic per rank at N = 8), and what the next forward needs travels back by
    all-gather: the bf16 shadow in perf mode (2 B/parameter: reduce-scatter + gather = 0.75 x the all-reduce's bytes), the fp32
    masters in parity mode.  Everything else (embeddings, MAG, biases, LayerNorms: 25 M parameters whose gradients every rank
    needs anyway or that are too small to matter) stays replicated.  In perf mode a rank's fp32 masters OUTSIDE its shard go
    stale; gather_masters() refreshes them for state_dict() / checkpoints.
    Unmeasured on hardware (no multi-GPU box for the builder): correctness is pinned by tests/test_dp_gpu.py (two ranks ==
    the replicated path bit for bit in fp32; replicas share one bf16 shadow in perf mode)."""

    def __init__(self, core, rank, world, group=None):
        self.core, self.rank, self.world, self.pg = core, rank, world, group
        lo, hi = int(core.sh_begin), int(core.sh_end)
        per = -(-(hi - lo) // world)
        per = (per + 255) // 256 * 256                   # shard boundaries on 1-KB marks (AdamW works on 16-byte quads)
        self.lo, self.hi, self.per = lo, hi, per
        self.bounds = [(min(hi, lo + r * per), min(hi, lo + (r + 1) * per)) for r in range(world)]
        self.a, self.b = self.bounds[rank]

    def split(self, ranges):
        """(pieces reduced to one owner [(off, n, owner)], pieces every rank needs [(off, n)]) of flat gradient ranges"""
        owned, shared = [], []
        for off, n in ranges:
            end, cur = off + n, off
            if end <= self.lo or off >= self.hi:
                shared.append((off, n))
                continue
            if cur < self.lo:
                shared.append((cur, self.lo - cur)); cur = self.lo
            for r, (a, b) in enumerate(self.bounds):
                x, y = max(cur, a), min(end, b)
                if y > x:
                    owned.append((x, y - x, r))
            if end > self.hi:
                shared.append((self.hi, end - self.hi))
        return owned, shared

    def reduce_to_owners(self, reducer, owned):
        """sum over the ranks of every owned piece, delivered to its owner only (the other ranks' copies are dead afterwards)"""
        g = reducer.g
        dst = lambda r: dist.get_global_rank(self.pg, r) if self.pg is not None else r

        def run():
            for off, n, r in owned:
                dist.reduce(g[off: off + n], dst=dst(r), op=dist.ReduceOp.SUM, group=self.pg)
        if reducer.cuda:
            ev = reducer._event()
            ev.record(torch.cuda.current_stream(g.device))
            with torch.cuda.stream(reducer.comm_stream):
                reducer.comm_stream.wait_event(ev)
                run()
        else:
            run()

    def gather_updated(self):
        """after the owners' updates: every rank gets the operands of the next forward -- the bf16 shadow (perf mode) or the fp32
        masters (parity mode) of all shards"""
        core = self.core
        t = core.shadow if core.compute_dtype == torch.bfloat16 else core.params
        self._gather([t[a:b] for a, b in self.bounds])

    def _gather(self, parts):
        if len({p.numel() for p in parts}) == 1:
            dist.all_gather(parts, parts[self.rank].clone(), group=self.pg)
        else:                                   # the last shard is shorter: one broadcast per shard
            for r, p in enumerate(parts):
                if p.numel():
                    dist.broadcast(p, src=dist.get_global_rank(self.pg, r) if self.pg is not None else r, group=self.pg)

    def gather_masters(self):
        """fp32 masters of every shard on every rank (state_dict(), checkpoints, evaluation in parity mode)"""
        self._gather([self.core.params[a:b] for a, b in self.bounds])

    def clear_dead_gradients(self):
        """the gradient copies of shards this rank does not own were never reduced here: zero them (optimizer.zero_grad())"""
        g = self.core.grads
        if self.a > self.lo:
            g[self.lo: self.a].zero_()
        if self.hi > self.b:
            g[self.b: self.hi].zero_()


class DataParallel(object):
    """Wraps a MAG_BertForSequenceClassification: hooks the engine's backward stages to the reducer."""

    def __init__(self, model, optimizer=None, process_group=None, row_capacity=0):
        """row_capacity: tokens per rank and step the row-wise word-embedding exchange must hold (train_batch_size * max_seq_length);
        0 = the size of the first single-call step (a later, larger batch then raises instead of silently diverging)"""
        self.model = model
        self.row_capacity = int(row_capacity)
        self.pg = process_group
        self.core = model._core
        # Wire format of the gradient exchange.  auto: bf16 only where the links are the bound -- bf16 perf mode, RCCL, and a
        # TWO-GPU group (xGMI is point-to-point: two GPUs share ONE link, so the 443 MB fp32 exchange lasts longer than the whole
        # backward; with 4 / 8 GPUs a ring spreads over 3 / 7 links and the fp32 exchange hides under the backward and the first
        # optimizer launch, while the two staging passes (fp32 -> bf16 -> fp32 over 91 M elements, 1.1 GB of HBM traffic on the comm
        # stream) cost 0.25 - 0.85 ms per step -- measured with a one-rank RCCL group, profiles/r04_dp_force.txt).
        wire = os.environ.get("MB_DP_GRAD_DTYPE", "auto")
        if wire == "auto":
            on_rccl = dist.is_initialized() and dist.get_backend(process_group) == "nccl"
            few_links = dist.is_initialized() and dist.get_world_size(process_group) == 2
            wire = "bf16" if (self.core.compute_dtype == torch.bfloat16 and self.core.grads.is_cuda and on_rccl and few_links) else "fp32"
        self.reducer = GradReducer(self.core.grads, process_group, torch.bfloat16 if wire == "bf16" else torch.float32)
        self.world = self.reducer.world
        self.plan, self.tail = stage_plan(self.core)
        self.core.stage_hooks.insert(0, self._on_stage)
        self.sync = True          # set False on gradient-accumulation micro-steps (multimodal_driver.py:383)
        self.optimizer = optimizer
        self.late_ranges = []
        self.split_last = os.environ.get("MB_DP_SPLIT_LAST", "1") != "0"
        # The word-embedding gradient (30,522 x 768 fp32 = 94 MB, 21 % of the payload, produced LAST) has at most B*L non-zero
        # rows: it is exchanged row-wise (exchange_embedding_rows).  MB_DP_SPARSE_EMB=0 keeps it in the dense last piece.
        self.word = None
        if os.environ.get("MB_DP_SPARSE_EMB", "1") != "0" and self.reducer.active:
            for name, off, numel, shape, decay in self.core.tensors:
                if name.endswith("word_embeddings.weight") or name.endswith("word_embedding.weight"):
                    self.word = (off, numel, tuple(shape))
        self.word_capacity = None           # rows per rank, agreed over the group at the first exchange
        self._micro_since_sync = 0
        # The step as ONE engine call with the exchange issued from C (mb_bert_train_step_dp / mb_xlnet_train_step_dp): what
        # train_step() runs whenever the step ends with the optimizer.  MB_DP_ENGINE=0 keeps every step on the stage-driven path
        # below (which also serves gradient-accumulation micro-steps and foreign optimizers).
        self.comm = None              # created at the first single-call step (the engine has its real size by then)
        self._last_fused = False
        self._comm_enabled = self.reducer.active and self.core.grads.is_cuda and os.environ.get("MB_DP_ENGINE", "1") != "0"
        # MB_DP_SHARD_OPT=1: the optimizer update of the layers' GEMM weights is sharded over the ranks (OptimizerShards: reduce to
        # the owner, update one shard, all-gather the operands).  Runs on the stage-driven path (the single engine call keeps the
        # replicated update).  Design + equality tests only: nothing about it has been measured on more than one GPU.
        # Round 5: with the single engine call available the sharding lives INSIDE it (csrc/comm.hip: reduce-scatter of every layer piece,
        # AdamW over this rank's slices, in-place all-gathers of the next forward's operands on the comm stream); the Python-driven
        # OptimizerShards below remains the MB_DP_ENGINE=0 form.
        self.shards = None
        self.shard_in_engine = False
        force1 = os.environ.get("MB_DP_SHARD_FORCE", "0") == "1"          # (one-rank group, identity collectives: one-GPU timing only)
        if self.reducer.active and (self.reducer.world > 1 or force1) and os.environ.get("MB_DP_SHARD_OPT", "0") == "1":
            if self._comm_enabled:
                self.shard_in_engine = True
            else:
                self.shards = OptimizerShards(self.core, dist.get_rank(process_group), self.reducer.world, process_group)
                self.core._dp_shards = self.shards  # (state_dict / sync_weights gather the stale masters first: ADVICE r4)
                self.word = None                    # (dense last piece: keeps this path's plan simple)
        # every rank draws its own dropout masks (the reference is single-process: nothing to be faithful to; identical masks on
        # every shard would correlate the regularisation noise).  The mixed seed is what get_rng_state() saves.
        if self.reducer.active and self.reducer.world > 1:
            rank = dist.get_rank(process_group)
            self.core.seed = (self.core.seed ^ ((0x9E3779B97F4A7C15 * (rank + 1)) & ((1 << 63) - 1))) & ((1 << 63) - 1)
        # exposed communication: timing events around the two places where the compute stream waits for the comm stream
        self._tev = [torch.cuda.Event(enable_timing=True) for _ in range(4)] if self.core.grads.is_cuda else None
        self._tev_used = [False, False]
        if optimizer is not None:
            optimizer.grad_scale = 1.0 / self.world
            optimizer._dp = self
        try:
            model._dp = self              # (train_step finds it on micro-steps, which carry no optimizer)
        except Exception:
            pass

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

    def _split_word(self, ranges):
        """ranges minus the word-embedding table (handled by the row exchange)"""
        if self.word is None:
            return list(ranges), False
        w0, wn, _ = self.word
        out, hit = [], False
        for off, n in ranges:
            if off <= w0 and w0 + wn <= off + n:
                hit = True
                if w0 > off:
                    out.append((off, w0 - off))
                if off + n > w0 + wn:
                    out.append((w0 + wn, off + n - (w0 + wn)))
            else:
                out.append((off, n))
        return out, hit

    def _exchange_word_rows(self):
        red = self.reducer
        w0, wn, shape = self.word
        ids = self.core.batch_ids()
        if self.word_capacity is None:          # one-time agreement on the per-rank row capacity (largest engine of the group)
            cap = torch.tensor([self.core.max_B * self.core.max_L], dtype=torch.int64, device=ids.device if red.cuda else "cpu")
            dist.all_reduce(cap, op=dist.ReduceOp.MAX, group=red.pg)
            self.word_capacity = int(cap.item())
        table = self.core.grads[w0: w0 + wn].view(shape)
        if red.cuda:
            ev = red._event()
            ev.record(torch.cuda.current_stream(self.core.grads.device))
            with torch.cuda.stream(red.comm_stream):
                red.comm_stream.wait_event(ev)
                exchange_embedding_rows(table, ids, self.word_capacity, red.pg)
        else:
            exchange_embedding_rows(table, ids, self.word_capacity, red.pg)

    def _on_stage(self, stage):
        if not self.sync:
            if stage == 0:
                self._micro_since_sync += 1
            return
        if self.shards is not None:
            owned, shared = self.shards.split(self.plan[stage] + (self.tail if stage == len(self.plan) - 1 else []))
            self.shards.reduce_to_owners(self.reducer, owned)
            self.reducer.reduce_ranges(shared)
            if stage == len(self.plan) - 1:
                self._micro_since_sync = 0
                self._timed_wait(0, lambda cs: self.reducer.wait())
            return
        if stage < len(self.plan) - 1:
            self.reducer.reduce_ranges(self.plan[stage])
            return
        # last stage: its pieces (embeddings + MAG: 94 MB, and the small tail) are produced last and would be fully exposed.
        # The compute stream only waits for everything BEFORE them; AdamW.step() updates the already-reduced ranges under this
        # last all-reduce and calls finish() before it touches the late ranges (split_last = False: plain full wait here).
        early = self.reducer.mark()
        # with accumulated micro-steps the touched rows are the union over the micro-steps: dense exchange for that step
        dense, sparse = self._split_word(self.plan[stage]) if self._micro_since_sync == 0 else (list(self.plan[stage]), False)
        self._micro_since_sync = 0
        self.reducer.reduce_ranges(dense)
        if sparse:
            self._exchange_word_rows()
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

    def fused_ready(self):
        """True when the next optimizer step can be the single engine call (mb_*_train_step_dp): a synchronising step with the C-side
        exchange available.  Micro-steps of a gradient-accumulation window ran as plain single calls before it (micro_ready); the
        step then exchanges the word-embedding table densely (Comm.set_row_exchange)."""
        return self._comm_enabled and self.sync

    def micro_ready(self):
        """True when a gradient-accumulation micro-step can be the plain single call (no exchange, no stage hooks)"""
        return self._comm_enabled and self.shards is None

    def get_comm(self, tokens):
        """the C-side exchange object, created collectively at the first single-call step"""
        if self.comm is None and self._comm_enabled:
            try:
                self.comm = Comm(self.core, self.pg, self.reducer.wire_dtype, sparse_rows=self.word is not None,
                                 row_capacity=max(self.row_capacity, int(tokens)))
                if self.shard_in_engine:
                    self.comm.set_sharding(True)
                self.core._dp_comm = self.comm
            except Exception as ex:         # no RCCL to load, communicator refused, ...: say so and keep training on the stage-driven path
                import sys
                print("warning: the single-call data-parallel step is unavailable (%s); the gradient exchange stays with the "
                      "stage-driven path (torch.distributed)" % (ex,), file=sys.stderr)
                self._comm_enabled = False
        return self.comm

    def exposed_ms(self):
        """Time the compute stream of the LAST step spent stalled on the gradient exchange (both waits: before the early
        AdamW ranges and before the late ones); synchronises.  0 when every piece had landed by the time it was needed."""
        if self._last_fused:
            return self.comm.exposed_ms()
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
