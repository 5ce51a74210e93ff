"""MAG_BertModel / MAG_BertForSequenceClassification -- drop-in surfaces of /root/reference/bert.py:76-324,
backed by the native MI355X step executor (csrc/engine.hip) instead of transformers' BertEmbeddings /
BertEncoder / BertPooler.

Kept verbatim from the reference: class names, constructor `(config, multimodal_config)`, forward argument order and
defaults, returned tuples `((loss,) logits, ...)` / `(sequence_output, pooled_output)`, `from_pretrained(name,
multimodal_config=..., num_labels=1)`, and every state-dict key (bert.embeddings.*, bert.encoder.layer.{i}.*,
bert.pooler.dense.*, bert.MAG.*, classifier.*) so reference / HuggingFace checkpoints load unchanged.

Different by design (MI355X-first):
  * all parameters are views of ONE flat fp32 buffer (weight-decay group first) with a matching flat gradient buffer:
    the optimizer is one fused launch and the data-parallel all-reduce runs on a handful of large contiguous ranges;
  * forward/backward are single C calls that enqueue the whole pass on the current HIP stream;
  * `compute_dtype=torch.bfloat16` (perf mode) keeps bf16 activations + a bf16 operand shadow of the GEMM weights;
    `torch.float32` (parity mode) runs exact-fp32 MFMA and matches the CPU reference logits to < 1e-3;
  * the optional arguments of forward are built on the HIP path too (head_mask, inputs_embeds, output_attentions,
    output_hidden_states, position_ids; encoder_hidden_states / encoder_attention_mask are ignored unless config.is_decoder, like the
    reference); a decoder configuration raises
    NotImplementedError instead of silently taking a slow path.
"""
import ctypes as C
import os

import torch
import torch.nn as nn

from . import _lib
from .global_configs import ACOUSTIC_DIM, VISUAL_DIM


class BertConfig(object):
    """The subset of transformers.BertConfig the path reads (bert-base-uncased defaults)."""

    def __init__(self, vocab_size=30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                 intermediate_size=3072, hidden_act="gelu", hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1,
                 max_position_embeddings=512, type_vocab_size=2, initializer_range=0.02, layer_norm_eps=1e-12,
                 pad_token_id=0, num_labels=1, **kwargs):
        if hidden_act != "gelu":
            raise NotImplementedError("only erf-GELU (bert-base-uncased) is built")
        self.vocab_size = vocab_size
        self.hidden_size = hidden_size
        self.num_hidden_layers = num_hidden_layers
        self.num_attention_heads = num_attention_heads
        self.intermediate_size = intermediate_size
        self.hidden_act = hidden_act
        self.hidden_dropout_prob = hidden_dropout_prob
        self.attention_probs_dropout_prob = attention_probs_dropout_prob
        self.max_position_embeddings = max_position_embeddings
        self.type_vocab_size = type_vocab_size
        self.initializer_range = initializer_range
        self.layer_norm_eps = layer_norm_eps
        self.pad_token_id = pad_token_id
        self.num_labels = num_labels
        self.output_attentions = False
        self.output_hidden_states = False
        self.is_decoder = False
        for k, v in kwargs.items():
            setattr(self, k, v)


class _Holder(nn.Module):
    """Plain container so parameters get the reference's dotted names."""


class _EngineFn(torch.autograd.Function):
    """logits = model(batch): the forward was already enqueued; backward hands dlogits to the engine, which
    accumulates into the flat gradient buffer (p.grad are views of it)."""

    @staticmethod
    def forward(ctx, anchor, logits, core, inputs_embeds=None):
        ctx.core = core
        ctx.want_emb = inputs_embeds is not None
        return logits.view_as(logits)

    @staticmethod
    def backward(ctx, dlogits):
        # autograd may already have accumulated other loss terms into the p.grad views: then this node adds to the buffer.  The
        # store path of the reference loop (model(...); loss.backward(); optimizer.step(); zero_grad()) survives when nothing on
        # the torch side has written into the flat gradient buffer since it was zeroed (_Core.torch_wrote_grads)
        if ctx.core.torch_wrote_grads():
            ctx.core.mark_grads_zero(False)
        ctx.core._backward(dlogits.contiguous().float())
        return torch.zeros((), device=dlogits.device), None, None, ctx.core.inputs_embeds_grad() if ctx.want_emb else None


class _BaseFn(torch.autograd.Function):
    """(sequence_output, pooled_output) = MAG_BertModel(batch) with an autograd edge (bert.py:233-237 returns autograd tensors):
    the forward was already enqueued; backward turns (d_sequence_output, d_pooled_output) into the engine's entry gradients and
    runs the encoder / MAG / embedding stages, accumulating into the flat gradient buffer (p.grad are views of it)."""

    @staticmethod
    def forward(ctx, anchor, seq, pooled, core, inputs_embeds=None):
        ctx.core = core
        ctx.want_emb = inputs_embeds is not None
        ctx.save_for_backward(pooled)
        return seq.view_as(seq), pooled.view_as(pooled)

    @staticmethod
    def backward(ctx, d_seq, d_pooled):
        core = ctx.core
        (pooled,) = ctx.saved_tensors
        cd = core.compute_dtype
        ds = None if d_seq is None else d_seq.to(cd).contiguous()
        dz = None if (d_pooled is None or core.kind != "bert") else (d_pooled.float() * (1.0 - pooled * pooled)).to(cd).contiguous()    # tanh'
        if core.torch_wrote_grads():          # as in _EngineFn: next to somebody else's gradients this node accumulates
            core.mark_grads_zero(False)
        core.backward_outputs(ds, dz)
        return torch.zeros((), device=core.device), None, None, None, core.inputs_embeds_grad() if ctx.want_emb else None


class _Core(object):
    """Flat parameter/gradient storage + engine handle shared by the model classes."""

    def __init__(self, config, multimodal_config, visual_dim, acoustic_dim, compute_dtype, device, kind="bert",
                 injection_index=1):
        self.kind = kind                     # "bert" (mb_bert_*) or "xlnet" (mb_xlnet_*)
        self.injection_index = injection_index
        if not torch.cuda.is_available():
            raise _lib.MagbertError("MAG-BERT runs on the HIP path only: no ROCm device visible (no CPU fallback)")
        self.lib = _lib.lib()
        self.config = config
        self.mc = multimodal_config
        self.V, self.A = int(visual_dim), int(acoustic_dim)
        self.compute_dtype = compute_dtype
        self.dt = _lib.DT_BF16 if compute_dtype == torch.bfloat16 else _lib.DT_F32
        self.device = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
        self.handle = None
        self.max_B, self.max_L = 0, 0
        self.seed = int(torch.initial_seed())
        self.step = 0
        self.training_last = False
        self._own_stream = None
        self.stage_hooks = []          # callables hook(stage) run after each backward stage (DataParallel)
        self._make_engine(1, 8)
        n = self._fn("param_count")(self.handle)
        self.n_params = n
        self.n_decay = self._fn("decay_count")(self.handle)
        # end of the flat range the optimizer updates: MAG-XLNet keeps a frozen slot (transformer.mask_emb) behind it
        self.n_update_end = n if kind == "bert" else self.lib.mb_xlnet_trainable_count(self.handle)
        b, e = C.c_size_t(), C.c_size_t()
        self._fn("shadow_range")(self.handle, C.byref(b), C.byref(e))
        self.sh_begin, self.sh_end = b.value, e.value
        self.params = torch.zeros(n, dtype=torch.float32, device=self.device)
        self.grads = torch.zeros(n, dtype=torch.float32, device=self.device)
        self.shadow = torch.zeros(n if self.dt == _lib.DT_BF16 else 1, dtype=torch.bfloat16, device=self.device)
        self.ws = None
        self.tensors = self._tensor_table()
        self.anchor = torch.zeros((), device=self.device, requires_grad=True)
        self.weights_dirty = True
        self.loss_buf = torch.zeros(2, dtype=torch.float32, device=self.device)   # [last step, running sum]
        self._optional = (None, None, None, None, None)
        self._gz = True             # the flat gradient buffer holds zeros (mirror of the engine's flag: survives a re-created engine)
        self._gz_version = self.grads._version

    # -- engine lifecycle ---------------------------------------------------------------------------
    def _fn(self, name):
        return getattr(self.lib, "mb_%s_%s" % (self.kind, name))

    @property
    def n_layers(self):
        return self.config.num_hidden_layers if self.kind == "bert" else self.config.n_layer

    def _cfg(self, B, L):
        c, mc = self.config, self.mc
        if self.kind == "xlnet":
            return _lib.XlnetEngineConfig(
                c.vocab_size, c.d_model, c.n_layer, c.n_head, c.d_inner, c.num_labels, self.V, self.A, int(self.injection_index),
                c.layer_norm_eps, 1e-5, float(mc.beta_shift), c.dropout, c.summary_last_dropout, float(mc.dropout_prob),
                self.dt, int(B), int(L))
        return _lib.BertEngineConfig(
            c.vocab_size, c.hidden_size, c.num_hidden_layers, c.num_attention_heads, c.intermediate_size,
            c.max_position_embeddings, c.type_vocab_size, c.num_labels, self.V, self.A, c.pad_token_id,
            c.layer_norm_eps, 1e-5, float(mc.beta_shift), c.hidden_dropout_prob, c.attention_probs_dropout_prob,
            float(mc.dropout_prob), self.dt, int(B), int(L))

    def _make_engine(self, B, L):
        h = C.c_void_p()
        cfg = self._cfg(B, L)
        _lib.check(self._fn("create")(C.byref(cfg), C.byref(h)))       # raises (and keeps the old engine) on a bad shape
        if self.handle is not None:
            self._fn("destroy")(self.handle)
            self.ws = None
        self.handle = h
        self._optional = (None, None, None, None, None)    # a new engine starts without head_mask / inputs_embeds / position_ids / perm_mask
        self.max_B, self.max_L = B, L

    def _comm_join(self):
        """sharded optimizer update under data parallel (distributed.Comm.set_sharding): the operands of this pass come back from the
        other ranks by all-gathers on the comm stream -- the current stream waits for the last of them (a no-op otherwise)"""
        comm = getattr(self, "_dp_comm", None)
        if comm is not None and comm.sharding and comm.handle is not None:
            _lib.check(self.lib.mb_comm_join(comm.handle, self.stream()))

    def refresh_sharded_state(self, adam=True):
        """sharded optimizer update: every rank holds current fp32 masters (and Adam moments) of ITS slices only; gather the rest
        before anything reads the whole flat buffers (state_dict, checkpoints, sync_weights).  Collective: every rank must call it."""
        old = getattr(self, "_dp_shards", None)          # the Python-driven form (distributed.OptimizerShards, MB_DP_ENGINE=0)
        if old is not None:
            old.gather_masters()
            if adam and hasattr(self, "_adam_m"):
                old._gather([self._adam_m[a:b] for a, b in old.bounds])
                old._gather([self._adam_v[a:b] for a, b in old.bounds])
            return
        comm = getattr(self, "_dp_comm", None)
        if comm is None or not comm.sharding or comm.handle is None:
            return
        self._comm_join()
        st = self.stream()
        _lib.check(self.lib.mb_comm_gather_shards(comm.handle, _lib.ptr(self.params), 4, st))
        if adam and hasattr(self, "_adam_m"):
            _lib.check(self.lib.mb_comm_gather_shards(comm.handle, _lib.ptr(self._adam_m), 4, st))
            _lib.check(self.lib.mb_comm_gather_shards(comm.handle, _lib.ptr(self._adam_v), 4, st))

    def _ensure(self, B, L, join=True):
        # join=False: the pass about to run is the data-parallel single call itself, which waits for the previous step's all-gathers
        # on its own -- piece by piece when the sharded update cut its forward (a full join here would undo that overlap)
        if join:
            self._comm_join()
        if B > self.max_B or L > self.max_L or self.ws is None:
            # "logically zero, physically stale" gradients are a fact only the OLD engine knows: make them real zeros before it goes
            self.materialize_grads()
            self._make_engine(max(B, self.max_B), max(L, self.max_L))
            nbytes = self._fn("workspace_bytes")(self.handle)
            self.ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            _lib.check(self._fn("bind")(self.handle, _lib.ptr(self.params), _lib.ptr(self.grads),
                                             _lib.ptr(self.shadow) if self.dt == _lib.DT_BF16 else None,
                                             _lib.ptr(self.ws), nbytes))
            self.weights_dirty = True
            _lib.check(self._fn("mark_grads_zero")(self.handle, 1 if self._gz else 0))

    def _tensor_table(self):
        out = []
        name = C.create_string_buffer(160)
        off, numel, ndim, decay = C.c_size_t(), C.c_size_t(), C.c_int(), C.c_int()
        shape = (C.c_int64 * 4)()
        for i in range(self._fn("num_tensors")(self.handle)):
            _lib.check(self._fn("tensor_info")(self.handle, i, name, 160, C.byref(off), C.byref(numel), C.byref(ndim),
                                                    shape, C.byref(decay)))
            out.append((name.value.decode(), off.value, numel.value, tuple(shape[k] for k in range(ndim.value)),
                        int(decay.value)))      # 1 decay, 0 no-decay, 2 frozen (never receives a gradient)
        return out

    def stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    class _Hop(object):
        """The engine forks/joins internal side streams with events.  On the legacy NULL stream those hand-offs were
        observed to be unreliable on ROCm 7.2 (intermittent stale gradients), so when the caller is on the default
        stream the pass hops onto a private non-default stream and joins back afterwards."""

        def __init__(self, core):
            self.core = core
            self.cur = torch.cuda.current_stream(core.device)
            self.hop = self.cur.cuda_stream == 0
            self.ctx = None

        def __enter__(self):
            if self.hop:
                if self.core._own_stream is None:
                    self.core._own_stream = torch.cuda.Stream(device=self.core.device)
                self.core._own_stream.wait_stream(self.cur)
                self.ctx = torch.cuda.stream(self.core._own_stream)
                self.ctx.__enter__()
            return self

        def __exit__(self, *exc):
            if self.hop:
                self.ctx.__exit__(*exc)
                self.cur.wait_stream(self.core._own_stream)
            return False

    def sync_weights(self):
        """refresh bf16 shadow + packed MAG operands from the fp32 masters (after load / manual edits)"""
        if self.dt == _lib.DT_BF16:
            self.refresh_sharded_state(adam=False)       # (sharded update: the shadow must not be rebuilt from stale masters)
        with _Core._Hop(self):
            _lib.check(self._fn("sync_weights")(self.handle, self.stream()))
        self.weights_dirty = False

    # -- passes --------------------------------------------------------------------------------------
    def _pinned_batch(self, tensors, B, L):
        """True when the batch can be read by the GPU where it is: six pinned, contiguous host tensors of the engine's dtypes
        (what prefetch.PinnedBatchRing yields).  Anything else goes through torch's .to(device)."""
        want = (torch.int64, torch.float32, torch.float32, torch.int64, torch.int64, torch.float32)
        shapes = ((B, L), (B, L, self.V), (B, L, self.A), (B, L), (B, L), None)
        for t, dt, shp in zip(tensors, want, shapes):
            if t is None:
                continue
            if t.is_cuda or t.dtype != dt or not t.is_contiguous() or (shp is not None and tuple(t.shape) != shp) or not t.is_pinned():
                return False
        return True

    def _inputs(self, input_ids, visual, acoustic, attention_mask, token_type_ids, labels, gather_in_step=False):
        """-> (pointers in the C ABI's order: ids, vis, aco, mask, seg, labels ; objects to keep alive).
        Pinned host batches are never copied by torch: mb_bert_train_step's prologue gathers them across PCIe itself
        (gather_in_step), the other passes get them through ONE gather launch (mb_bert_load_batch) into the engine's staging
        buffers.  Device / pageable tensors take the reference's route, `t.to(DEVICE)` (multimodal_driver.py:359)."""
        B, L = input_ids.shape
        dev = self.device
        lab_in = None if labels is None else labels.reshape(-1)
        nl = int(self.config.num_labels)
        if lab_in is not None and lab_in.numel() != B * (1 if nl > 1 else nl):
            raise ValueError("labels: expected %d values for a batch of %d, got %d" % (B, B, lab_in.numel()))
        if lab_in is not None and nl > 1:
            # fused cross entropy (head.hip): class indices travel as fp32; what torch's CrossEntropyLoss would reject or treat
            # specially must not silently become class 0 (ignore_index = -100 is not built)
            lf = lab_in.detach().float()
            if bool(((lf != lf.round()) | (lf < 0) | (lf >= nl)).any()):
                raise ValueError("labels of the fused cross entropy must be class indices in [0, %d) (ignore_index is not supported)" % nl)
        six = (input_ids, visual, acoustic, attention_mask, token_type_ids, lab_in)
        if self._pinned_batch(six, B, L):
            ptrs = [None if t is None else C.c_void_p(t.data_ptr()) for t in six]
            if gather_in_step:
                return ptrs, six
            staged = (C.c_void_p * 6)()
            with _Core._Hop(self):
                _lib.check(self._fn("load_batch")(self.handle, ptrs[0], ptrs[1], ptrs[2], ptrs[3], ptrs[4], ptrs[5], B, L, staged,
                                                   self.stream()))
            off = staged[0] - self.ws.data_ptr()
            self._ids_dev = self.ws[off: off + B * L * 8].view(torch.int64)          # the engine's staging copy of input_ids
            return [C.c_void_p(staged[i]) if staged[i] else None for i in range(6)], six
        nb = not input_ids.is_cuda and input_ids.is_pinned()      # pinned host tensors: asynchronous copies on this stream
        ids = input_ids.to(dev, torch.int64, non_blocking=nb).contiguous()
        msk = attention_mask.to(dev, torch.int64, non_blocking=nb).contiguous()
        seg = token_type_ids.to(dev, torch.int64, non_blocking=nb).contiguous()
        vis = visual.to(dev, torch.float32, non_blocking=nb).contiguous()
        aco = acoustic.to(dev, torch.float32, non_blocking=nb).contiguous()
        if vis.shape != (B, L, self.V) or aco.shape != (B, L, self.A):
            raise ValueError("visual/acoustic must be [B, L, %d] / [B, L, %d], got %s / %s" %
                             (self.V, self.A, tuple(vis.shape), tuple(aco.shape)))
        lab = None if labels is None else labels.to(dev, torch.float32, non_blocking=nb).contiguous().view(-1)
        keep = (ids, vis, aco, msk, seg, lab)
        self._ids_dev = ids.reshape(-1)
        return [_lib.ptr(t) for t in keep], keep

    def _set_optional(self, head_mask=None, inputs_embeds=None, position_ids=None, perm=None, mems=None):
        """head_mask [n_layers][n_heads] / inputs_embeds [B*L][H] (fp32 device tensors) / position_ids [B*L] (int64, MAG-BERT) /
        perm [B][L][L] (uint8, MAG-XLNet: who may not attend to whom) or None -> engine state; sticky in the engine, so every pass
        states what it wants (the single-call step refuses to run with any of them set)."""
        want = (head_mask, inputs_embeds, position_ids, perm, mems)
        if all(x is None for x in want) and all(x is None for x in self._optional):
            return
        if self.kind != "bert":
            if position_ids is not None:
                raise NotImplementedError("position_ids is an argument of MAG-BERT only (XLNet has relative positions)")
            _lib.check(self.lib.mb_xlnet_set_head_mask(self.handle, _lib.ptr(head_mask)))
            _lib.check(self.lib.mb_xlnet_set_inputs_embeds(self.handle, _lib.ptr(inputs_embeds)))
            _lib.check(self.lib.mb_xlnet_set_perm_mask(self.handle, _lib.ptr(perm)))
            # mems: [n_layer][B][mlen][H] in the activation dtype (xlnet.py:374-385; include/magbert_hip.h: mb_xlnet_set_mems)
            _lib.check(self.lib.mb_xlnet_set_mems(self.handle, _lib.ptr(mems), 0 if mems is None else int(mems.shape[2])))
        else:
            if perm is not None or mems is not None:
                raise NotImplementedError("perm_mask / mems are arguments of MAG-XLNet only")
            _lib.check(self.lib.mb_bert_set_head_mask(self.handle, _lib.ptr(head_mask)))
            _lib.check(self.lib.mb_bert_set_inputs_embeds(self.handle, _lib.ptr(inputs_embeds)))
            _lib.check(self.lib.mb_bert_set_position_ids(self.handle, _lib.ptr(position_ids)))
        self._optional = want          # kept alive: the engine holds raw pointers through the backward

    def mark_grads_zero(self, known_zero=True):
        """tells the engine the flat gradient buffer holds zeros (it then stores, instead of accumulating, the layer weight
        gradients of the next backward -- include/magbert_hip.h: mb_bert_mark_grads_zero).  Called by everything of ours that
        zeroes the buffer; code that writes into `.grad` tensors by hand between a zero_grad() and a backward must pass False."""
        if not known_zero:
            self.materialize_grads()          # whatever the caller is about to add to: real zeros where a fused step skipped them
        self._gz = bool(known_zero)
        self._gz_version = self.grads._version     # torch-side writes into the buffer (or any `.grad` view of it) move this counter
        if self.handle is not None:
            _lib.check(self._fn("mark_grads_zero")(self.handle, 1 if known_zero else 0))

    def torch_wrote_grads(self):
        """True when something on the torch side (autograd accumulating another loss term into a `.grad` view, a hand edit) has
        written into the flat gradient buffer since our own zeroing declared it known-zero: every `.grad` is a view of the one
        buffer and shares its version counter, which the engine's own raw-pointer writes never touch.  A second backward without a
        zero_grad() in between finds the flag already consumed (_gz False) and accumulates as well."""
        return (not self._gz) or self.grads._version != getattr(self, "_gz_version", -1)

    def materialize_grads(self):
        """A single-call step that ends with the optimizer does not write the zeros of optimizer.zero_grad() over the layers'
        GEMM weight gradients (the next backward overwrites them: include/magbert_hip.h, mb_bert_materialize_grads).  Everything
        that is about to read the flat gradient buffer calls this first; a no-op when nothing is stale."""
        if self.handle is not None and self.ws is not None and self._fn("grads_stale")(self.handle):
            with _Core._Hop(self):
                _lib.check(self._fn("materialize_grads")(self.handle, self.stream()))

    def head_mask_table(self, head_mask):
        """transformers get_head_mask (bert.py:206-207): [n_heads] (every layer) or [n_layers][n_heads] (or the broadcast
        5-D form of it) -> fp32 [n_layers][n_heads] on the device.  Masks that differ per sample or per position are not built."""
        if head_mask is None:
            return None
        NL, nh = self.n_layers, (self.config.num_attention_heads if self.kind == "bert" else self.config.n_head)
        hm = torch.as_tensor(head_mask).to(self.device, torch.float32)
        if hm.dim() == 1 and hm.numel() == nh:
            hm = hm[None].expand(NL, nh)
        elif hm.numel() == NL * nh and (hm.dim() == 2 or tuple(hm.shape) == (NL, 1, nh, 1, 1)):
            hm = hm.reshape(NL, nh)
        else:
            raise NotImplementedError("head_mask must be [num_heads] or [num_layers, num_heads], got %s" % (tuple(hm.shape),))
        return hm.contiguous()

    def inputs_embeds_grad(self):
        """gradient of the inputs_embeds of the last forward, after its backward: fp32 [B, L, H] (a copy)"""
        p = self._fn("inputs_embeds_grad")(self.handle)
        B, L, H = self._emb_shape
        off = p - self.ws.data_ptr()
        return self.ws[off: off + B * L * H * 4].view(torch.float32).view(B, L, H).clone()

    def forward(self, input_ids, visual, acoustic, attention_mask, token_type_ids, labels, training, mems=None, head_mask=None,
                inputs_embeds=None, position_ids=None, perm=None):
        dev = self.device
        if inputs_embeds is not None:               # bert.py:158-168 / xlnet.py:306-313: shapes come from the embeddings, the ids are not read
            inputs_embeds = inputs_embeds.detach().to(dev, torch.float32).contiguous()
            B, L, H = inputs_embeds.shape
            if H != self.config.hidden_size:
                raise ValueError("inputs_embeds must be [B, L, %d]" % self.config.hidden_size)
            self._emb_shape = (B, L, H)
            input_ids = torch.zeros(B, L, dtype=torch.int64, device=dev)
        B, L = input_ids.shape
        self._ensure(B, L)
        if self.weights_dirty:
            self.sync_weights()
        if position_ids is not None:
            position_ids = position_ids.to(dev, torch.int64).expand(B, L).contiguous()
        if perm is not None:
            perm = perm.to(dev, torch.uint8).contiguous()
            if tuple(perm.shape) != (B, L, L):
                raise ValueError("perm_mask must be [B, L, L] = %s, got %s" % ((B, L, L), tuple(perm.shape)))
        self._set_optional(self.head_mask_table(head_mask), inputs_embeds, position_ids, perm, mems)
        ptr, keep = self._inputs(input_ids, visual, acoustic, attention_mask, token_type_ids, labels)
        logits = torch.empty(B, self.config.num_labels, dtype=torch.float32, device=dev)
        if training:
            self.step += 1
        self._keep = (keep, logits)                 # the engine keeps raw pointers until the backward
        self._lab_ptr = ptr[5]
        self.training_last = bool(training)
        with _Core._Hop(self):
            _lib.check(self._fn("forward")(self.handle, ptr[0], ptr[1], ptr[2], ptr[3], ptr[4], ptr[5], B, L, 1 if training else 0,
                                           self.seed & (2 ** 64 - 1), self.step, _lib.ptr(logits), C.c_void_p(self.loss_buf.data_ptr()),
                                           C.c_void_p(self.loss_buf.data_ptr() + 4) if ptr[5] is not None else None,
                                           self.stream()))
        return logits

    def _backward(self, dlogits=None, loss_scale=1.0):
        nstage = self.n_layers + 2
        lab = self._lab_ptr
        if dlogits is None and lab is None:
            raise ValueError("fused backward needs the labels passed to forward()")
        with _Core._Hop(self):
            self._gz = False                    # (the engine consumed its flag at stage 0)
            for s in range(nstage):
                _lib.check(self._fn("backward")(self.handle, _lib.ptr(dlogits), lab if dlogits is None else None,
                                                float(loss_scale), s, s + 1, self.stream()))
                for hook in self.stage_hooks:
                    hook(s)

    def batch_ids(self):
        """device tensor of the token ids of the last forward (data parallel: the rows of the word-embedding gradient)"""
        return self._ids_dev

    def fused_step_blocker(self):
        """why one optimizer step cannot be ONE engine call (mb_bert_train_step), or None"""
        if self.stage_hooks:
            return "backward stage hooks are installed (data parallel: the gradient exchange is issued between stages)"
        if self.kind == "xlnet" and os.environ.get("MB_OVERLAP_WGRAD", "0") not in ("", "0"):
            return "MB_OVERLAP_WGRAD=1: the side-stream weight gradients of MAG-XLNet are driven stage by stage"
        return None

    def train_step(self, input_ids, visual, acoustic, attention_mask, token_type_ids, labels, opt, loss_scale=1.0, mode=2, comm=None):
        """One optimizer step as ONE engine call (include/magbert_hip.h: mb_bert_train_step): the step prologue (batch gather,
        dropout keys, AdamW scalars -> device memory) followed by every kernel of the step, launched one by one (mode 2) or as
        a replayed hipGraph (mode 1).  opt: None (gradient-accumulation micro-step, no update) or AdamW.flat_step_args().
        comm (distributed.Comm): the data-parallel form, mb_*_train_step_dp -- the same step as a chain of graphs with the
        gradient exchange issued from C between them."""
        B, L = input_ids.shape
        self._ensure(B, L, join=comm is None)
        if self.weights_dirty:
            self.sync_weights()
        if labels is None:
            raise ValueError("the fused step needs label_ids")
        self._set_optional(None, None, None)
        ptr, keep = self._inputs(input_ids, visual, acoustic, attention_mask, token_type_ids, labels, gather_in_step=True)
        if not hasattr(self, "_logit_bufs"):
            self._logit_bufs = {}
        logits = self._logit_bufs.get(B)
        if logits is None:          # one persistent buffer per batch size: its address is part of a captured graph
            logits = self._logit_bufs[B] = torch.empty(B, self.config.num_labels, dtype=torch.float32, device=self.device)
        self.step += 1
        self._keep = (keep, logits)
        self._lab_ptr = None        # the engine's own staging copy: a later stand-alone backward needs a new forward
        self.training_last = True
        o = opt or {}
        if comm is not None and opt is None:
            raise ValueError("the data-parallel single-call step ends with the optimizer (micro-steps exchange nothing)")
        extra = () if comm is None else (comm.handle,)
        with _Core._Hop(self):
            _lib.check(self._fn("train_step" if comm is None else "train_step_dp")(
                self.handle, ptr[0], ptr[1], ptr[2], ptr[3], ptr[4], ptr[5], B, L,
                self.seed & (2 ** 64 - 1), self.step, _lib.ptr(logits), C.c_void_p(self.loss_buf.data_ptr()),
                C.c_void_p(self.loss_buf.data_ptr() + 4), _lib.ptr(o.get("m")), _lib.ptr(o.get("v")), o.get("lr", 0.0),
                o.get("beta1", 0.9), o.get("beta2", 0.999), o.get("eps", 1e-6), o.get("weight_decay", 0.0), int(o.get("t", 1)),
                1 if o.get("correct_bias", True) else 0, o.get("grad_scale", 1.0), float(loss_scale), int(mode), self.stream(), *extra))
        self._gz = opt is not None          # the fused AdamW left the gradients zeroed / a micro-step left them populated
        self._gz_version = self.grads._version
        return logits

    def stage_step(self, input_ids, visual, acoustic, attention_mask, token_type_ids, labels, loss_scale=1.0, mode=1):
        """forward + MSE + backward of one training step cut at the backward stages (include/magbert_hip.h: mb_bert_stage_forward /
        mb_bert_stage_backward): every pass is ONE replayed graph, the stage hooks (data parallel: the gradient exchange) run
        between them.  What a data-parallel rank runs instead of ~210 kernel launches from Python."""
        B, L = input_ids.shape
        self._ensure(B, L)
        if self.weights_dirty:
            self.sync_weights()
        self._set_optional(None, None, None)
        ptr, keep = self._inputs(input_ids, visual, acoustic, attention_mask, token_type_ids, labels, gather_in_step=True)
        if not hasattr(self, "_logit_bufs"):
            self._logit_bufs = {}
        logits = self._logit_bufs.get(B)
        if logits is None:
            logits = self._logit_bufs[B] = torch.empty(B, self.config.num_labels, dtype=torch.float32, device=self.device)
        self.step += 1
        self._keep = (keep, logits)
        self._lab_ptr = None
        self.training_last = True
        off = self.lib.mb_bert_staged_input_ids(self.handle) - self.ws.data_ptr()
        self._ids_dev = self.ws[off: off + B * L * 8].view(torch.int64)          # the engine's staging copy of input_ids
        with _Core._Hop(self):
            _lib.check(self.lib.mb_bert_stage_forward(
                self.handle, ptr[0], ptr[1], ptr[2], ptr[3], ptr[4], ptr[5], B, L, self.seed & (2 ** 64 - 1), self.step, _lib.ptr(logits),
                C.c_void_p(self.loss_buf.data_ptr()), C.c_void_p(self.loss_buf.data_ptr() + 4), int(mode), self.stream()))
            self._gz = False
            for s in range(self.n_layers + 2):
                _lib.check(self.lib.mb_bert_stage_backward(self.handle, float(loss_scale), s, int(mode), self.stream()))
                for hook in self.stage_hooks:
                    hook(s)
        return logits

    def graph_stats(self):
        cap, rep = C.c_size_t(), C.c_size_t()
        _lib.check(self._fn("graph_stats")(self.handle, C.byref(cap), C.byref(rep)))
        return cap.value, rep.value

    def _act(self, ptr, B, L):
        """fp32 copy of a [B*L][H] activation of the last forward (workspace pointer -> tensor)"""
        H = self.config.hidden_size
        es = 2 if self.dt == _lib.DT_BF16 else 4
        off = ptr - self.ws.data_ptr()
        return self.ws[off: off + B * L * H * es].view(self.compute_dtype).view(B, L, H).float()

    def hidden_states(self, B, L):
        """all_hidden_states of the last forward: n_layers + 1 tensors [B, L, H] (bert.py:227-237 / xlnet.py:363-392)"""
        return tuple(self._act(self._fn("hidden_state")(self.handle, i), B, L) for i in range(self.n_layers + 1))

    def xl_attentions(self, B, L, training):
        """MAG-XLNet output_attentions (xlnet.py:387-427): n_layers tensors [B, n_head, L, L] fp32, the attention probabilities
        after dropout -- the probabilities the forward saved for its backward times, in train mode, the counter-hash dropout mask
        of the layer's site (regenerated on the host: an optional debugging output, not a hot path)."""
        from . import rng
        nh, p = self.config.n_head, float(self.config.dropout)
        es = 2 if self.dt == _lib.DT_BF16 else 4
        out = []
        for l in range(self.n_layers):
            lp = C.c_int()
            ptr = self.lib.mb_xlnet_attention_probs(self.handle, l, C.byref(lp))
            if not ptr:
                raise _lib.MagbertError("no forward has run")
            LP = lp.value
            off = ptr - self.ws.data_ptr()
            a = self.ws[off: off + B * nh * LP * LP * es].view(self.compute_dtype).view(B, nh, LP, LP)[:, :, :L, :L].float()
            if training and p > 0.0:
                key = rng.make_key(self.seed, self.step, rng.XS_LAYER0 + 8 * l, p)
                a = a * torch.from_numpy(rng.keep_mult(B * nh * L * L, key)).view(B, nh, L, L).to(a.device)
            if self._optional[0] is not None:            # head_mask: attn_prob * head_mask (xlnet.py:383)
                a = a * self._optional[0][l].view(1, nh, 1, 1)
            out.append(a.contiguous())
        return tuple(out)

    def attention_buffer(self, B, L):
        """arms the next forward to write every layer's attention probabilities (MAG-BERT); returns the buffer"""
        if self.kind != "bert":
            raise NotImplementedError("output_attentions is built for MAG-BERT only")
        nh = self.config.num_attention_heads
        buf = torch.empty(self.n_layers, B, nh, L, L, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.mb_bert_set_attention_output(self.handle, _lib.ptr(buf)))
        return buf

    def attention_done(self):
        if self.kind == "bert":
            _lib.check(self.lib.mb_bert_set_attention_output(self.handle, None))

    def backward_outputs(self, d_seq, d_pre):
        """backward of the base model from the gradients of its outputs -- MAG-BERT: (sequence_output, pooler pre-activation),
        mb_bert_backward_outputs; MAG-XLNet: the dropped last hidden state, mb_xlnet_backward_outputs -- then the encoder / MAG /
        embedding stages"""
        nstage = self.n_layers + 2
        self._gz = False
        with _Core._Hop(self):
            if self.kind == "bert":
                _lib.check(self.lib.mb_bert_backward_outputs(self.handle, _lib.ptr(d_seq), _lib.ptr(d_pre), self.stream()))
            else:
                _lib.check(self.lib.mb_xlnet_backward_outputs(self.handle, _lib.ptr(d_seq), self.stream()))
            for hook in self.stage_hooks:
                hook(0)
            for s in range(1, nstage):
                _lib.check(self._fn("backward")(self.handle, None, None, 1.0, s, s + 1, self.stream()))
                for hook in self.stage_hooks:
                    hook(s)

    def xl_model_output(self, B, L):
        """MAG_XLNetModel's return value (xlnet.py:396-405): the last layer's output after the final dropout, fp32 [B, L, H]"""
        with _Core._Hop(self):
            p = self.lib.mb_xlnet_model_output(self.handle, self.stream())
        if not p:
            raise _lib.MagbertError("no forward has run")
        return self._act(p, B, L).clone()          # the engine's scratch is reused by the backward: hand out a copy

    def xl_query_stream(self, target_mapping, B, L):
        """MAG-XLNet's query stream (xlnet.py:238-240, 306-313, 374-399; include/magbert_hip.h: mb_xlnet_query_stream) as a post-pass
        over the eval forward that just ran: target_mapping [B, M, L] -> (logits_g [B, num_labels], hidden_g: n_layers + 1 fp32
        tensors [B, M, H], the last one = output_g)."""
        if self.kind != "xlnet":
            raise NotImplementedError("target_mapping is an argument of MAG-XLNet only")
        tm = target_mapping.detach().to(self.device, torch.float32).contiguous()
        if tm.dim() != 3 or tm.shape[0] != B or tm.shape[2] != L:
            raise ValueError("target_mapping must be [batch, num_predict, seq_len] = [%d, M, %d], got %s" % (B, L, tuple(tm.shape)))
        M, H = int(tm.shape[1]), self.config.hidden_size
        need = self.lib.mb_xlnet_query_stream_scratch_bytes(self.handle, B, M, L)
        scratch = torch.empty(need + 256, dtype=torch.uint8, device=self.device)
        base = (-scratch.data_ptr()) % 256
        logits = torch.empty(B, self.config.num_labels, dtype=torch.float32, device=self.device)
        with _Core._Hop(self):
            _lib.check(self.lib.mb_xlnet_query_stream(self.handle, _lib.ptr(tm), M, C.c_void_p(scratch.data_ptr() + base), need,
                                                      _lib.ptr(logits), self.stream()))
        stride = self.lib.mb_xlnet_query_stream_state_bytes(self.handle, B, M)
        es = 2 if self.dt == _lib.DT_BF16 else 4
        states = tuple(scratch[base + i * stride: base + i * stride + B * M * H * es].view(self.compute_dtype).view(B, M, H).float()
                       for i in range(self.n_layers + 1))
        return logits, states

    def sequence_output(self, B, L):
        return self._act(self._fn("sequence_output")(self.handle), B, L)

    def pooled_output(self, B):
        H = self.config.hidden_size
        p = self.lib.mb_bert_pooled_output(self.handle)
        off = p - self.ws.data_ptr()
        return self.ws[off: off + B * H * 4].view(torch.float32).view(B, H).clone()

    def stage_ranges(self, stage):
        offs, lens = (C.c_size_t * 8)(), (C.c_size_t * 8)()
        n = self._fn("stage_grad_ranges")(self.handle, stage, offs, lens, 8)
        return [(offs[i], lens[i]) for i in range(n)]

    def __del__(self):
        try:
            if self.handle is not None:
                self._fn("destroy")(self.handle)
        except Exception:
            pass


def _attach_parameters(root, core, prefix_filter=None, strip=""):
    """Registers one nn.Parameter per reference tensor, as a view of the flat buffer, under its dotted name."""
    for name, off, numel, shape, decay in core.tensors:
        if prefix_filter is not None and not name.startswith(prefix_filter):
            continue
        rel = name[len(strip):] if strip and name.startswith(strip) else name
        parts = rel.split(".")
        mod = root
        for part in parts[:-1]:
            if not hasattr(mod, part):
                mod.add_module(part, _Holder())
            mod = getattr(mod, part)
        p = nn.Parameter(core.params[off: off + numel].view(shape))
        if decay == 2:          # frozen (e.g. XLNet mask_emb): no gradient ever -> HF AdamW would skip it; so do we
            p._mb_flat = None
        else:
            p.grad = core.grads[off: off + numel].view(shape)
            p._mb_flat = (core, off, numel, decay)
        mod.register_parameter(parts[-1], p)


def _init_weights(core):
    """transformers PreTrainedModel._init_weights law (bert.py:90,249): Linear/Embedding ~ N(0, initializer_range),
    biases 0, LayerNorm 1/0, embedding padding row 0 -- also applied to MAG's Linears, as in the reference where
    init_weights() runs after self.MAG is built."""
    std = core.config.initializer_range
    g = torch.Generator(device=core.device)
    g.manual_seed(int(torch.initial_seed()) & 0x7FFFFFFF)
    with torch.no_grad():
        for name, off, numel, shape, decay in core.tensors:
            v = core.params[off: off + numel]
            leaf = name.split(".")[-1]
            if "LayerNorm" in name or "layer_norm" in name:
                v.fill_(1.0 if leaf == "weight" else 0.0)
            elif leaf == "bias":
                v.zero_()
            else:
                v.normal_(0.0, std, generator=g)
                if name.endswith("word_embeddings.weight"):       # BERT: nn.Embedding(padding_idx=pad_token_id)
                    H = shape[1]
                    v[core.config.pad_token_id * H:(core.config.pad_token_id + 1) * H].zero_()
    core.weights_dirty = True


def load_pretrained_state_dict(model, state_dict, base_prefix, source="checkpoint"):
    """The key handling of transformers 3.0.2 PreTrainedModel.from_pretrained (what multimodal_driver.py:317-323 relies on):
      * legacy LayerNorm names: `gamma` -> `weight`, `beta` -> `bias` (the published bert-base-uncased file uses them);
      * base-model prefix: a checkpoint saved from the bare BertModel / XLNetModel loads into `model.<base_prefix>`; a
        checkpoint saved from a model with a head loads into the bare model with the prefix stripped;
      * `*.position_ids` buffers of newer transformers are ignored;
      * everything of the checkpoint that has no home here is reported as unexpected (e.g. `cls.predictions.*`), every
        parameter of the model the checkpoint does not provide as missing (it keeps its fresh init: `bert.MAG.*`,
        `classifier.*`, ...); a tensor with the wrong shape is an error, as in transformers.
    Nothing is dropped silently: both lists are logged (logging.WARNING, same wording as transformers) and returned."""
    import logging
    log = logging.getLogger(__name__)
    sd = {}
    for k, v in state_dict.items():
        nk = k.replace("gamma", "weight") if "gamma" in k else k
        nk = nk.replace("beta", "bias") if "beta" in nk else nk
        if nk.endswith("position_ids"):
            continue
        sd[nk] = v
    own = dict(model.state_dict())
    pre = base_prefix + "."
    model_has = any(k.startswith(pre) for k in own)
    ckpt_has = any(k.startswith(pre) for k in sd)
    if model_has and not ckpt_has:
        sd = {pre + k: v for k, v in sd.items()}
    elif not model_has and ckpt_has:
        sd = {(k[len(pre):] if k.startswith(pre) else k): v for k, v in sd.items()}
    missing = [k for k in own if k not in sd]
    unexpected = [k for k in sd if k not in own]
    errors = ["size mismatch for %s: copying a param with shape %s from checkpoint, the shape in current model is %s." %
              (k, tuple(sd[k].shape), tuple(own[k].shape)) for k in own if k in sd and tuple(sd[k].shape) != tuple(own[k].shape)]
    if errors:
        raise RuntimeError("Error(s) in loading state_dict for %s:\n\t%s" % (model.__class__.__name__, "\n\t".join(errors)))
    nn.Module.load_state_dict(model, {k: v for k, v in sd.items() if k in own}, strict=False)
    model._core.weights_dirty = True
    name = model.__class__.__name__
    if unexpected:
        log.warning("Some weights of the model checkpoint at %s were not used when initializing %s: %s", source, name, unexpected)
    if missing:
        log.warning("Some weights of %s were not initialized from the model checkpoint at %s and are newly initialized: %s",
                    name, source, missing)
    return {"missing_keys": missing, "unexpected_keys": unexpected, "error_msgs": []}


def _pretrained_file(path, cls_name):
    if os.path.isdir(path):
        path = os.path.join(path, "pytorch_model.bin")
    if not os.path.isfile(path):
        raise OSError("from_pretrained(%r): no local checkpoint (offline build; pass a directory holding pytorch_model.bin or a "
                      "state-dict file, or construct %s(config, multimodal_config) and load_state_dict yourself)" % (path, cls_name))
    return path


class _MagBertBase(nn.Module):
    def _front(self, input_ids, inputs_embeds, attention_mask, token_type_ids, position_ids):
        """argument checks and defaults of MAG_BertModel.forward (bert.py:158-177) -> (B, L, attention_mask, token_type_ids)"""
        if input_ids is not None and inputs_embeds is not None:
            raise ValueError("You cannot specify both input_ids and inputs_embeds at the same time")
        if input_ids is None and inputs_embeds is None:
            raise ValueError("You have to specify either input_ids or inputs_embeds")
        B, L = input_ids.shape if input_ids is not None else inputs_embeds.shape[:-1]
        dev = self._core.device
        if position_ids is not None:      # bert.py:211-216 -> BertEmbeddings: rows of the position table, [B, L] or broadcastable to it
            position_ids = torch.as_tensor(position_ids)
            if position_ids.shape[-1] != L or position_ids.numel() not in (L, B * L):
                raise ValueError("position_ids must be [batch, seq_len] or [1, seq_len], got %s" % (tuple(position_ids.shape),))
            position_ids = position_ids.reshape(-1, L)
            if int(position_ids.min()) < 0 or int(position_ids.max()) >= self.config.max_position_embeddings:
                raise IndexError("position_ids out of range of the %d-row position table" % self.config.max_position_embeddings)
            if torch.equal(position_ids.cpu(), torch.arange(L).expand(position_ids.shape[0], L)):
                position_ids = None       # the default: the kernels index the table with the row's place in its sample
        self._position_ids = position_ids
        if attention_mask is None:
            attention_mask = torch.ones(B, L, dtype=torch.int64, device=dev)       # bert.py:173-174
        if token_type_ids is None:
            token_type_ids = torch.zeros(B, L, dtype=torch.int64, device=dev)      # bert.py:175-177
        return B, L, attention_mask, token_type_ids

    def _unsupported(self, **kw):
        for k, v in kw.items():
            if v is not None and v is not False:
                raise NotImplementedError("%s is not supported by the HIP path (unused by multimodal_driver.py:363-370)" % k)

    def to(self, *args, **kwargs):
        # parameters are views of one flat HBM buffer created on the target device; the reference's
        # model.to(DEVICE) (multimodal_driver.py:325) is therefore a no-op
        dev = None
        for a in args:
            if isinstance(a, (torch.device, str)):
                dev = torch.device(a)
        dev = kwargs.get("device", dev)
        if dev is not None and torch.device(dev).type != "cuda":
            raise _lib.MagbertError("the HIP model cannot be moved off the ROCm device (no CPU fallback)")
        return self

    def cuda(self, device=None):
        return self

    def _load_from_state_dict(self, *a, **k):
        super()._load_from_state_dict(*a, **k)
        self._core.weights_dirty = True

    def load_state_dict(self, state_dict, strict=True, **kw):
        sd = {k: v for k, v in state_dict.items() if not k.endswith("position_ids")}
        r = super().load_state_dict(sd, strict=strict, **kw)
        self._core.weights_dirty = True
        return r

    def state_dict(self, *args, **kwargs):
        # sharded optimizer update under data parallel: a rank's fp32 masters outside its slices are stale until gathered
        self._core.refresh_sharded_state()
        return super().state_dict(*args, **kwargs)

    def init_weights(self):
        _init_weights(self._core)

    def get_rng_state(self):
        """(seed, step) of the counter-hash dropout: every mask is a pure function of (seed, step, site, element), so a
        resumed run that restores this pair draws exactly the masks the uninterrupted run would have drawn."""
        return {"seed": int(self._core.seed), "step": int(self._core.step)}

    def set_rng_state(self, state):
        self._core.seed, self._core.step = int(state["seed"]), int(state["step"])

    def stream_scope(self):
        """Context manager for a training / evaluation loop: when the caller sits on the legacy NULL stream, the whole
        loop (engine passes, AdamW, H2D copies) runs on the model's private stream instead of hopping NULL -> private ->
        NULL around every pass.  Each hop is a cross-queue dependency (measured: ~25 us between forward and backward,
        ~90 us between AdamW and the next forward)."""
        import contextlib
        core = self._core

        @contextlib.contextmanager
        def scope():
            cur = torch.cuda.current_stream(core.device)
            if cur.cuda_stream != 0:
                yield
                return
            if core._own_stream is None:
                core._own_stream = torch.cuda.Stream(device=core.device)
            core._own_stream.wait_stream(cur)
            with torch.cuda.stream(core._own_stream):
                yield
            cur.wait_stream(core._own_stream)
        return scope()

    def zero_grad(self, set_to_none=False):
        # p.grad are views of the flat gradient buffer the engine accumulates into: clear it in place
        self._core.grads.zero_()
        self._core.mark_grads_zero(True)

    def sync_weights(self):
        """call after editing parameters in place (bf16 mode keeps an operand shadow of the GEMM weights)"""
        self._core.sync_weights()

    base_model_prefix = "bert"

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, *model_args, config=None, **kwargs):
        """multimodal_driver.py:317-319.  There is no network here: the argument must be a local directory (or file) holding
        a `pytorch_model.bin`-style state dict, e.g. the published bert-base-uncased file.  Keys are mapped the way
        transformers 3.0.2 maps them (load_pretrained_state_dict: legacy gamma / beta names, base-model prefix); parameters
        the checkpoint lacks (bert.MAG.*, classifier.*) keep their fresh init exactly like in the reference, and both the
        missing and the unused keys are logged and kept in `model.loading_info` (`output_loading_info=True` returns them)."""
        multimodal_config = kwargs.pop("multimodal_config", model_args[0] if model_args else None)
        num_labels = kwargs.pop("num_labels", 1)
        want_info = kwargs.pop("output_loading_info", False)
        config = config or cls._default_config(num_labels)
        config.num_labels = num_labels
        model = cls(config, multimodal_config, **kwargs)
        path = _pretrained_file(pretrained_model_name_or_path, cls.__name__)
        sd = torch.load(path, map_location="cpu")
        model.loading_info = load_pretrained_state_dict(model, sd, cls.base_model_prefix, source=path)
        return (model, model.loading_info) if want_info else model

    @staticmethod
    def _default_config(num_labels):
        return BertConfig(num_labels=num_labels)


class _FusedStep(object):
    """Fused training surface shared by MAG_BertForSequenceClassification and MAG_XLNetForSequenceClassification
    (what the bundled driver's train_epoch and bench.py call instead of the reference's five-line step)."""

    def training_step(self, input_ids, visual, acoustic, attention_mask, token_type_ids, label_ids, loss_scale=1.0):
        """forward + loss + backward in two C calls, no host sync.  The loss is the one the reference's forward computes when it is
        given labels (bert.py:313-322 / xlnet.py:515-524): MSE for num_labels == 1 (what the driver trains, multimodal_driver.py:372-373),
        cross entropy over the class indices in label_ids otherwise.
        Returns the device scalar holding this step's loss (running sum is in .loss_running())."""
        core = self._core
        # data parallel, MB_DP_GRAPH=1: one replayed graph per pass / backward stage instead of kernel launches from Python.  Opt-in:
        # measured on one GPU (1-rank RCCL group, profiles/r03_dp_force.txt) the graphs are not faster (4.52 vs 4.43 ms per step; the
        # single-call step: 3.84) -- what the stage-driven step pays is the exchange machinery itself, not the launches
        if core.kind == "bert" and core.stage_hooks and os.environ.get("MB_DP_GRAPH", "0") == "1" and os.environ.get("MB_OVERLAP_WGRAD", "0") in ("", "0"):
            core.stage_step(input_ids, visual, acoustic, attention_mask, token_type_ids, label_ids, loss_scale=loss_scale)
            return core.loss_buf[0]
        core.forward(input_ids, visual, acoustic, attention_mask, token_type_ids, label_ids, True)
        core._backward(None, loss_scale)
        return core.loss_buf[0]

    def train_step(self, input_ids, visual, acoustic, attention_mask, token_type_ids, label_ids, optimizer=None,
                   loss_scale=1.0, graph=None):
        """One iteration of train_epoch's loop body (multimodal_driver.py:359-386): batch staging, forward, MSE, backward
        and -- when `optimizer` is given -- optimizer.step() + optimizer.zero_grad().  scheduler.step() stays with the caller.

        Where it can (either model, the driver's two parameter groups on this model's flat buffer) the whole iteration is ONE
        engine call, mb_*_train_step: a step prologue that gathers the batch (straight from pinned host memory if that is where
        it is) and puts this step's dropout keys / lr / bias correction into device memory, then every kernel of the step as ONE
        replayed hipGraph (the step is a single-stream kernel sequence: replay costs ~12 us of host time and runs as fast as the
        stream launches).  Under data parallel (optimizer._dp set by distributed.DataParallel) the same call becomes
        mb_*_train_step_dp: a chain of linear graphs with the gradient exchange issued from C between them (distributed.Comm).
        Gradient-accumulation micro-steps of a data-parallel rank (optimizer=None between two synchronising steps) are the plain
        single call without exchange and optimizer; the step that ends the window exchanges the accumulated gradients.
        graph="launches" (or MB_STEP_GRAPH=0) keeps the single call but launches the kernels one by one.  Otherwise (foreign
        optimizers, MB_DP_ENGINE=0) the passes are driven from here: training_step + optimizer.step(); graph=False forces that
        path, graph=True raises if the single call is unavailable.
        Pass optimizer=None on gradient-accumulation micro-steps.  Returns the device loss scalar."""
        core = self._core
        dp = getattr(optimizer, "_dp", None) if optimizer is not None else None
        mdp = getattr(self, "_dp", None)
        if optimizer is None and mdp is not None and not mdp.sync and mdp.micro_ready() and graph is not False and \
                core.kind in ("bert", "xlnet") and os.environ.get("MB_OVERLAP_WGRAD", "0") in ("", "0"):
            # gradient-accumulation micro-step of a data-parallel rank (multimodal_driver.py:375-376, 383): nothing is exchanged, so it is the
            # plain single call without the optimizer (the backward accumulates); the exchange of the step that ends the window
            # moves the word-embedding table densely (distributed.DataParallel._micro_since_sync)
            launches = graph == "launches" or (graph is None and os.environ.get("MB_STEP_GRAPH", "1") == "0")
            core.train_step(input_ids, visual, acoustic, attention_mask, token_type_ids, label_ids, None, loss_scale=loss_scale,
                            mode=2 if launches else 1)
            mdp._micro_since_sync += 1
            mdp._last_fused = True
            return core.loss_buf[0]
        if dp is not None and graph is not False and dp.fused_ready() and core.kind in ("bert", "xlnet") and \
                os.environ.get("MB_OVERLAP_WGRAD", "0") in ("", "0"):
            # data parallel: the same single call with the gradient exchange inside (mb_*_train_step_dp, distributed.Comm)
            opt = optimizer.flat_step_args(core, allow_dp=True)
            comm = None
            if opt is not None:
                B_, L_ = input_ids.shape
                core._ensure(B_, L_, join=False)
                comm = dp.get_comm(B_ * L_)
            if comm is not None:
                optimizer._t += 1
                opt["t"] = optimizer._t
                optimizer._opt_called = True
                launches = graph == "launches" or (graph is None and os.environ.get("MB_STEP_GRAPH", "1") == "0")
                comm.set_row_exchange(dp._micro_since_sync == 0)     # after micro-steps: the union of their rows -> dense table
                core.train_step(input_ids, visual, acoustic, attention_mask, token_type_ids, label_ids, opt, loss_scale=loss_scale,
                                mode=2 if launches else 1, comm=comm)
                dp._micro_since_sync = 0
                dp._last_fused = True
                return core.loss_buf[0]
        if dp is not None:
            dp._last_fused = False
        why = core.fused_step_blocker()
        opt = None
        if why is None and optimizer is not None:
            opt = optimizer.flat_step_args(core) if hasattr(optimizer, "flat_step_args") else None
            if opt is None:
                why = "the optimizer is not the two-group AdamW over this model's flat buffer"
        if graph is True and why is not None:
            raise _lib.MagbertError("single-call step unavailable: " + why)
        if why is not None or graph is False:
            self.training_step(input_ids, visual, acoustic, attention_mask, token_type_ids, label_ids, loss_scale=loss_scale)
            if optimizer is not None:
                optimizer.step()
                optimizer.zero_grad()
            return core.loss_buf[0]
        if optimizer is not None:
            optimizer._t += 1
            opt["t"] = optimizer._t
            optimizer._opt_called = True          # the update happens inside the step: lr schedulers see an optimizer step
        launches = graph == "launches" or (graph is None and os.environ.get("MB_STEP_GRAPH", "1") == "0")
        core.train_step(input_ids, visual, acoustic, attention_mask, token_type_ids, label_ids, opt, loss_scale=loss_scale,
                        mode=2 if launches else 1)
        return core.loss_buf[0]

    def eval_step(self, input_ids, visual, acoustic, attention_mask, token_type_ids, label_ids):
        """forward in the module's current mode with the fused MSE of eval_epoch (multimodal_driver.py:405-411): the batch
        loss is added to loss_running() on the device.  Returns the logits."""
        return self._core.forward(input_ids, visual, acoustic, attention_mask, token_type_ids, label_ids, self.training)

    def loss_running(self, reset=False):
        v = self._core.loss_buf[1].clone()
        if reset:
            self._core.loss_buf[1].zero_()
        return v

    # flat views for the fused optimizer / data parallel -------------------------------------------------
    @property
    def flat_params(self):
        return self._core.params

    @property
    def flat_grads(self):
        self._core.materialize_grads()
        return self._core.grads

    def materialize_grads(self):
        """after a fused train_step with an optimizer the `.grad` views of the encoder's GEMM weights hold stale values instead of
        the zeros of optimizer.zero_grad() until something needs them (_Core.materialize_grads); call this before reading them"""
        self._core.materialize_grads()


class MAG_BertModel(_MagBertBase):
    """bert.py:76-237.  forward -> (sequence_output, pooled_output).  Outputs are fp32 copies of engine activations
    (not differentiable; the trainable surface is MAG_BertForSequenceClassification, which is what the driver uses)."""

    def __init__(self, config, multimodal_config, visual_dim=VISUAL_DIM, acoustic_dim=ACOUSTIC_DIM,
                 compute_dtype=torch.float32, device=None, _core=None):
        super().__init__()
        self.config = config
        own = _core is None
        self._core = _core or _Core(config, multimodal_config, visual_dim, acoustic_dim, compute_dtype, device)
        _attach_parameters(self, self._core, prefix_filter="bert.", strip="bert.")
        if own:
            self.init_weights()

    def get_input_embeddings(self):
        return self.embeddings.word_embeddings

    def forward(self, input_ids, visual, acoustic, attention_mask=None, token_type_ids=None, position_ids=None,
                head_mask=None, inputs_embeds=None, encoder_hidden_states=None, encoder_attention_mask=None,
                output_attentions=None, output_hidden_states=None):
        """-> (sequence_output, pooled_output, (hidden_states), (attentions)) like bert.py:233-237.  sequence_output and
        pooled_output carry an autograd edge into the engine (a head built on top of this model trains the whole stack, and
        inputs_embeds receives its gradient); hidden_states / attentions are detached fp32 copies.  head_mask: [num_heads] or
        [num_layers, num_heads]; position_ids: rows of the position table, [B, L] or [1, L].  The decoder arguments
        (encoder_hidden_states / encoder_attention_mask) behave as in the reference: with config.is_decoder False -- every
        MAG-BERT configuration -- bert.py:185-201 sets the cross-attention mask to None and transformers 3.0.2's BertLayer only
        looks at encoder_hidden_states `if self.is_decoder`: the arguments are accepted and IGNORED; a decoder stack
        (config.is_decoder True: cross-attention layers the MAG encoder never builds) raises."""
        if getattr(self.config, "is_decoder", False):
            self._unsupported(is_decoder=True)
        output_attentions = output_attentions if output_attentions is not None else getattr(self.config, "output_attentions", False)
        output_hidden_states = (output_hidden_states if output_hidden_states is not None
                                else getattr(self.config, "output_hidden_states", False))
        B, L, attention_mask, token_type_ids = self._front(input_ids, inputs_embeds, attention_mask, token_type_ids, position_ids)
        core = self._core
        probs = None
        if output_attentions:
            core._ensure(B, L)
            probs = core.attention_buffer(B, L)
        try:
            core.forward(input_ids, visual, acoustic, attention_mask, token_type_ids, None, self.training, head_mask=head_mask,
                         inputs_embeds=inputs_embeds, position_ids=self._position_ids)
        finally:
            if output_attentions:
                core.attention_done()
        seq, pooled = core.sequence_output(B, L), core.pooled_output(B)
        if torch.is_grad_enabled():
            emb_edge = inputs_embeds if inputs_embeds is not None and inputs_embeds.requires_grad else None
            seq, pooled = _BaseFn.apply(core.anchor, seq, pooled, core, emb_edge)
        outputs = (seq, pooled)
        if output_hidden_states:
            outputs = outputs + (core.hidden_states(B, L),)
        if output_attentions:
            outputs = outputs + (tuple(probs[l] for l in range(core.n_layers)),)
        return outputs


class MAG_BertForSequenceClassification(_FusedStep, _MagBertBase):
    """bert.py:240-324."""

    def __init__(self, config, multimodal_config, visual_dim=VISUAL_DIM, acoustic_dim=ACOUSTIC_DIM,
                 compute_dtype=torch.float32, device=None):
        super().__init__()
        self.config = config
        self.num_labels = config.num_labels
        self._core = _Core(config, multimodal_config, visual_dim, acoustic_dim, compute_dtype, device)
        self.bert = MAG_BertModel(config, multimodal_config, visual_dim, acoustic_dim, compute_dtype, device, _core=self._core)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)        # bert.py:246 (p lives in the engine config)
        _attach_parameters(self, self._core, prefix_filter="classifier.")
        self.init_weights()

    # reference API ------------------------------------------------------------------------------------
    def forward(self, input_ids, visual, acoustic, attention_mask=None, token_type_ids=None, position_ids=None,
                head_mask=None, inputs_embeds=None, labels=None, output_attentions=None, output_hidden_states=None):
        output_attentions = output_attentions if output_attentions is not None else getattr(self.config, "output_attentions", False)
        output_hidden_states = (output_hidden_states if output_hidden_states is not None
                                else getattr(self.config, "output_hidden_states", False))
        B, L, attention_mask, token_type_ids = self._front(input_ids, inputs_embeds, attention_mask, token_type_ids, position_ids)
        core = self._core
        probs = None
        if output_attentions:
            core._ensure(B, L)
            probs = core.attention_buffer(B, L)
        try:
            logits = core.forward(input_ids, visual, acoustic, attention_mask, token_type_ids, None, self.training,
                                  head_mask=head_mask, inputs_embeds=inputs_embeds, position_ids=self._position_ids)
        finally:
            if output_attentions:
                core.attention_done()
        if torch.is_grad_enabled():
            emb_edge = inputs_embeds if inputs_embeds is not None and inputs_embeds.requires_grad else None
            logits = _EngineFn.apply(core.anchor, logits, core, emb_edge)
        outputs = (logits,)                                           # bert.py:309-311: (logits,) + outputs[2:]
        if output_hidden_states:
            outputs = outputs + (core.hidden_states(B, L),)
        if output_attentions:
            outputs = outputs + (tuple(probs[l] for l in range(core.n_layers)),)
        if labels is not None:                                        # bert.py:313-322
            if self.num_labels == 1:
                loss = torch.nn.functional.mse_loss(logits.view(-1), labels.to(logits.device).float().view(-1))
            else:
                loss = torch.nn.functional.cross_entropy(logits.view(-1, self.num_labels), labels.to(logits.device).view(-1))
            outputs = (loss,) + outputs
        return outputs
