"""MAG_XLNetModel / MAG_XLNetForSequenceClassification -- drop-in surfaces of /root/reference/xlnet.py:15-527 on the native
MI355X step executor (csrc/xlnet_engine.hip, csrc/xlnet_attention.hip).

Same class names, constructor `(config, multimodal_config)`, forward argument order and state-dict keys as the reference
(transformer.word_embedding.weight, transformer.mask_emb, transformer.layer.{i}.rel_attn.{q,k,v,o,r,r_r_bias,r_s_bias,
r_w_bias,seg_embed,layer_norm.*}, transformer.layer.{i}.ff.{layer_norm,layer_1,layer_2}.*, transformer.MAG.*,
sequence_summary.summary.*, logits_proj.*).  Built for the configuration the reference driver runs
(multimodal_driver.py:363-370: attention_mask + token_type_ids; perm_mask / input_mask are built too (forward kernel + its
adjoint); mems (forward and backward of MAG_XLNetForSequenceClassification; the base model under no_grad) and the new_mems return
(config.mem_len, use_cache); target_mapping (the query stream g, xlnet.py:306-313, 374-399) for inference -- eval mode under
torch.no_grad(), without mems / output_attentions; inputs_embeds, output_hidden_states / output_attentions (served from the activations the engine keeps for
its backward) and head_mask (scales each head's attention output inside the kernels) are built, and MAG_XLNetModel's output is
differentiable), sequence length <= 128, MAG injected in front of layer
XLNET_INJECTION_INDEX (global_configs.py:19, xlnet.py:371-372).
"""
import torch
import torch.nn as nn

from . import _lib
from .bert import _BaseFn, _Core, _EngineFn, _FusedStep, _MagBertBase, _attach_parameters, _init_weights
from .global_configs import ACOUSTIC_DIM, VISUAL_DIM, XLNET_INJECTION_INDEX


class XLNetConfig(object):
    """The subset of transformers.XLNetConfig (xlnet-base-cased) the path reads."""

    def __init__(self, vocab_size=32000, d_model=768, n_layer=12, n_head=12, d_inner=3072, ff_activation="gelu",
                 attn_type="bi", initializer_range=0.02, layer_norm_eps=1e-12, dropout=0.1, mem_len=None, reuse_len=None,
                 bi_data=False, clamp_len=-1, same_length=False, summary_type="last", summary_use_proj=True,
                 summary_activation="tanh", summary_last_dropout=0.1, num_labels=1, **kwargs):
        if ff_activation != "gelu" or attn_type != "bi" or bi_data or clamp_len != -1 or reuse_len not in (None, 0) \
                or summary_type != "last" or not summary_use_proj or summary_activation != "tanh":
            raise NotImplementedError("only the xlnet-base-cased configuration used by multimodal_driver.py is built")
        self.mem_len = mem_len          # > 0 with use_cache: forward() also returns new_mems (xlnet.py:81-91, 363-365, 406-407)
        self.vocab_size = vocab_size
        self.d_model = d_model
        self.hidden_size = d_model
        self.n_layer = n_layer
        self.n_head = n_head
        self.d_head = d_model // n_head
        self.d_inner = d_inner
        self.initializer_range = initializer_range
        self.layer_norm_eps = layer_norm_eps
        self.dropout = dropout
        self.summary_last_dropout = summary_last_dropout
        self.num_labels = num_labels
        self.mem_len = mem_len
        for k, v in kwargs.items():
            setattr(self, k, v)


class _XlBase(_MagBertBase):
    base_model_prefix = "transformer"          # from_pretrained (bert._MagBertBase) maps keys exactly like transformers 3.0.2

    @staticmethod
    def _default_config(num_labels):
        return XLNetConfig(num_labels=num_labels)


class MAG_XLNetModel(_XlBase):
    """xlnet.py:15-429.  forward -> (output [B, L, d_model],) as an fp32 copy of the engine activation."""

    def __init__(self, config, multimodal_config, visual_dim=VISUAL_DIM, acoustic_dim=ACOUSTIC_DIM,
                 compute_dtype=torch.float32, device=None, injection_index=XLNET_INJECTION_INDEX, _core=None):
        super().__init__()
        self.config = config
        own = _core is None
        self._core = _core or _Core(config, multimodal_config, visual_dim, acoustic_dim, compute_dtype, device, kind="xlnet",
                                    injection_index=injection_index)
        _attach_parameters(self, self._core, prefix_filter="transformer.", strip="transformer.")
        if own:
            self.init_weights()

    def get_input_embeddings(self):
        return self.word_embedding

    def forward(self, input_ids, visual, acoustic, attention_mask=None, mems=None, perm_mask=None, target_mapping=None,
                token_type_ids=None, input_mask=None, head_mask=None, inputs_embeds=None, use_cache=True,
                output_attentions=None, output_hidden_states=None):
        """-> (output [B, L, d_model], (hidden_states), (attentions)) like xlnet.py:400-429.  `output` is the last layer's
        hidden state after the final dropout (xlnet.py:396) and carries an autograd edge into the engine: a head built on this
        model trains the whole stack, and inputs_embeds receives its gradient."""
        output_attentions = output_attentions if output_attentions is not None else getattr(self.config, "output_attentions", False)
        output_hidden_states = (output_hidden_states if output_hidden_states is not None
                                else getattr(self.config, "output_hidden_states", False))
        B, L, attention_mask, token_type_ids, perm = _xl_front(self, input_ids, inputs_embeds, attention_mask, token_type_ids, input_mask, perm_mask)
        core = self._core
        mlen, stack = 0, None
        if mems is not None:
            mlen, input_ids, inputs_embeds, visual, acoustic, attention_mask, token_type_ids, perm, stack = _xl_mems_front(
                self, mems, input_ids, inputs_embeds, visual, acoustic, attention_mask, token_type_ids, perm)
        K = mlen + L
        core.forward(input_ids, visual, acoustic, attention_mask, token_type_ids, None, self.training, mems=stack, head_mask=head_mask,
                     inputs_embeds=inputs_embeds, perm=perm)
        if target_mapping is not None:         # xlnet.py:396-399: the query stream's output [B, M, d_model] is what the model returns
            _, gs = _xl_query_stream(self, target_mapping, B, L, mems, output_attentions)
            outputs = (gs[-1],)
            if output_hidden_states:
                outputs = outputs + (_xl_interleave(core.hidden_states(B, L), gs),)
            return outputs
        out = core.xl_model_output(B, K)[:, mlen:]
        if torch.is_grad_enabled():
            emb_edge = inputs_embeds if inputs_embeds is not None and inputs_embeds.requires_grad else None
            out, _ = _BaseFn.apply(core.anchor, out, out.new_zeros(1), core, emb_edge)
        outputs = (out,)
        mem_len = getattr(self.config, "mem_len", None)
        if mem_len is not None and mem_len > 0 and use_cache is True:          # xlnet.py:363-365, 406-407
            outputs = outputs + (_xl_new_mems(core, mems, mlen, B, K, mem_len),)
        if output_hidden_states:               # xlnet.py:363-392: the input of every layer (before the MAG injection) + the last output
            outputs = outputs + (tuple(h[:, mlen:] for h in core.hidden_states(B, K)),)
        if output_attentions:                  # xlnet.py:387-427: per layer [B, n_head, qlen, klen], after the attention dropout
            outputs = outputs + (tuple(a[:, :, mlen:] for a in core.xl_attentions(B, K, self.training)),)
        return outputs

def _xl_mems_front(model, mems, input_ids, inputs_embeds, visual, acoustic, attention_mask, token_type_ids, perm, allow_grad=False):
    """mems (xlnet.py:244-245, 276-293, 317-323, 374-385): n_layer tensors [mlen, B, d_model], the hidden states cached from the previous
    segment.  Keys / values of layer l run over cat([mems[l], h]); queries over h.  The engine takes the segment as klen = mlen + L
    rows per sample whose first mlen rows are placeholders -- dummy ids, zero modalities, visible (the reference's mems_mask is all
    zeros), segment id 0 (the reference's mem_pad) -- and replaces those rows of every layer's input by mems[l]
    (include/magbert_hip.h: mb_xlnet_set_mems); its backward clears those rows' gradients at every layer seam (the memories are
    detached, xlnet.py:91).  -> (mlen, the extended arguments..., stacked memories)"""
    core = model._core
    if torch.is_grad_enabled() and not allow_grad:
        raise NotImplementedError("the differentiable base model does not take mems (its output gradient would have to be mapped back "
                                  "onto klen rows): call it under torch.no_grad(), or train through MAG_XLNetForSequenceClassification")
    mems = list(mems)
    if len(mems) != core.n_layers or any(m is None for m in mems):
        raise ValueError("mems must hold one tensor per layer (%d), got %d" % (core.n_layers, len(mems)))
    B, L = input_ids.shape if input_ids is not None else inputs_embeds.shape[:2]
    mlen, H, dev = int(mems[0].shape[0]), core.config.d_model, core.device
    if any(tuple(m.shape) != (mlen, B, H) for m in mems):
        raise ValueError("every element of mems must be [mlen, batch, d_model] = %s" % ((mlen, B, H),))
    if mlen + L > 128:
        raise NotImplementedError("klen = mlen + seq_len = %d exceeds the 128 rows the relative-attention kernels hold" % (mlen + L))
    stack = torch.stack([m.detach().to(dev, torch.float32).permute(1, 0, 2) for m in mems]).to(core.compute_dtype).contiguous()
    pad = lambda t, fill, dt: torch.cat([torch.full((B, mlen) + tuple(t.shape[2:]), fill, dtype=dt, device=dev), t.to(dev, dt)], dim=1)
    if input_ids is not None:
        input_ids = pad(input_ids, 0, torch.int64)
    else:
        inputs_embeds = pad(inputs_embeds, 0.0, torch.float32)
    visual, acoustic = pad(visual, 0.0, torch.float32), pad(acoustic, 0.0, torch.float32)
    attention_mask = pad(attention_mask, 1, torch.int64)
    token_type_ids = pad(token_type_ids, 0, torch.int64)
    if perm is not None:                 # [B, L, L] -> [B, klen, klen]: every memory key is visible to every query
        ext = torch.zeros(B, mlen + L, mlen + L, dtype=torch.uint8, device=dev)
        ext[:, mlen:, mlen:] = perm.to(dev)
        perm = ext
    return mlen, input_ids, inputs_embeds, visual, acoustic, attention_mask, token_type_ids, perm, stack


def _xl_new_mems(core, mems, mlen, B, klen, mem_len):
    """xlnet.py:81-91 (reuse_len None), 363-365: per layer, the last mem_len rows of cat([prev_mem, curr_out]) where curr_out is the
    hidden state in front of the layer (before the MAG injection), [len, B, d_model], detached"""
    hs = core.hidden_states(B, klen)
    out = []
    for i in range(core.n_layers):
        cur = hs[i][:, mlen:].permute(1, 0, 2)
        new = cur if mems is None else torch.cat([mems[i].to(cur.device, cur.dtype), cur], dim=0)
        out.append(new[-mem_len:].detach().contiguous())
    return tuple(out)


def _xl_query_stream(model, target_mapping, B, L, mems, output_attentions):
    """target_mapping [B, M, L] -> the query stream g (xlnet.py:238-240, 306-313, 374-399), run by the engine as a post-pass over the
    forward that just finished (csrc/xlnet_engine.hip: mb_xlnet_query_stream).  The reference driver never passes it
    (multimodal_driver.py:363-370), so it is built for inference: eval mode under torch.no_grad(), no mems, no attention
    probabilities of the g stream.  -> (logits_g [B, num_labels], hidden_g: n_layer + 1 tensors [B, M, d_model])"""
    if model.training or torch.is_grad_enabled():
        raise NotImplementedError("target_mapping (the query stream) is built for inference: call model.eval() and run under "
                                  "torch.no_grad()")
    if mems is not None:
        raise NotImplementedError("target_mapping together with mems")
    if output_attentions:
        raise NotImplementedError("output_attentions together with target_mapping (the g stream's probabilities are not kept)")
    return model._core.xl_query_stream(target_mapping, B, L)


def _xl_interleave(hs, gs):
    """xlnet.py:412-416: with a query stream every hidden_states entry is the pair (h, g), flattened in layer order"""
    return tuple(t for pair in zip(hs, gs) for t in pair)


def _xl_front(model, input_ids, inputs_embeds, attention_mask, token_type_ids, input_mask=None, perm_mask=None):
    """argument checks and defaults of MAG_XLNetModel.forward (xlnet.py:201-213, 255-286).  Returns (B, L, attention_mask,
    token_type_ids, perm): `perm` is None, or the uint8 [B, L, L] form of the reference's data_mask > 0 (xlnet.py:258-286:
    data_mask[i, j, b] = input_mask[j, b] + perm_mask[i, j, b]; query i may not attend to key j, i == j exempt) -- in which case the
    attention_mask handed to the engine is all ones, the byte mask carries everything."""
    if input_ids is not None and inputs_embeds is not None:
        raise ValueError("You cannot specify both input_ids and inputs_embeds at the same time")          # xlnet.py:201-203
    if input_ids is None and inputs_embeds is None:
        raise ValueError("You have to specify either input_ids or inputs_embeds")                          # xlnet.py:211-213
    B, L = input_ids.shape if input_ids is not None else inputs_embeds.shape[:-1]
    dev = model._core.device
    # xlnet.py:258-262
    assert input_mask is None or attention_mask is None, \
        "You can only use one of input_mask (uses 1 for padding) or attention_mask (uses 0 for padding, added for compatbility with BERT). Please choose one."
    perm = None
    if perm_mask is not None or (input_mask is not None and bool(((input_mask != 0) & (input_mask != 1)).any())):
        im = input_mask if input_mask is not None else (None if attention_mask is None else 1.0 - attention_mask.float())     # xlnet.py:263-264
        data = perm_mask.to(dev, torch.float32) if perm_mask is not None else torch.zeros(B, L, L, device=dev)
        if im is not None:
            data = data + im.to(dev, torch.float32)[:, None, :]              # xlnet.py:265-266: input_mask[None] + perm_mask, as [B, i, j]
        perm = (data > 0).to(torch.uint8)                                    # xlnet.py:283-284
        attention_mask = torch.ones(B, L, dtype=torch.int64, device=dev)
    elif input_mask is not None:                                             # a 0/1 input_mask is attention_mask negated (xlnet.py:263-264)
        attention_mask = (input_mask.to(dev) <= 0).to(torch.int64)
    if attention_mask is None:
        attention_mask = torch.ones(B, L, dtype=torch.int64, device=dev)
    if token_type_ids is None:
        token_type_ids = torch.zeros(B, L, dtype=torch.int64, device=dev)
    return B, L, attention_mask, token_type_ids, perm


class MAG_XLNetForSequenceClassification(_FusedStep, _XlBase):
    """xlnet.py:432-527."""

    def __init__(self, config, multimodal_config, visual_dim=VISUAL_DIM, acoustic_dim=ACOUSTIC_DIM,
                 compute_dtype=torch.float32, device=None, injection_index=XLNET_INJECTION_INDEX):
        super().__init__()
        self.config = config
        self.num_labels = config.num_labels
        self._core = _Core(config, multimodal_config, visual_dim, acoustic_dim, compute_dtype, device, kind="xlnet",
                           injection_index=injection_index)
        self.transformer = MAG_XLNetModel(config, multimodal_config, visual_dim, acoustic_dim, compute_dtype, device,
                                          injection_index, _core=self._core)
        _attach_parameters(self, self._core, prefix_filter="sequence_summary.")
        _attach_parameters(self, self._core, prefix_filter="logits_proj.")
        self.init_weights()

    def forward(self, input_ids, visual, acoustic, attention_mask=None, mems=None, perm_mask=None, target_mapping=None,
                token_type_ids=None, input_mask=None, head_mask=None, inputs_embeds=None, use_cache=True, labels=None,
                output_attentions=None, output_hidden_states=None):
        output_attentions = output_attentions if output_attentions is not None else getattr(self.config, "output_attentions", False)
        output_hidden_states = (output_hidden_states if output_hidden_states is not None
                                else getattr(self.config, "output_hidden_states", False))
        B, L, attention_mask, token_type_ids, perm = _xl_front(self, input_ids, inputs_embeds, attention_mask, token_type_ids, input_mask, perm_mask)
        core = self._core
        mlen, stack = 0, None
        if mems is not None:
            if inputs_embeds is not None and inputs_embeds.requires_grad and torch.is_grad_enabled():
                raise NotImplementedError("inputs_embeds with a gradient together with mems")
            mlen, input_ids, inputs_embeds, visual, acoustic, attention_mask, token_type_ids, perm, stack = _xl_mems_front(
                self, mems, input_ids, inputs_embeds, visual, acoustic, attention_mask, token_type_ids, perm, allow_grad=True)
        K = mlen + L
        logits = core.forward(input_ids, visual, acoustic, attention_mask, token_type_ids, None, self.training, mems=stack, head_mask=head_mask,
                              inputs_embeds=inputs_embeds, perm=perm)
        gs = None
        if target_mapping is not None:         # xlnet.py:396-399, 506-509: the head summarises output_g (its last target row)
            logits, gs = _xl_query_stream(self, target_mapping, B, L, mems, output_attentions)
        if torch.is_grad_enabled():
            emb_edge = inputs_embeds if inputs_embeds is not None and inputs_embeds.requires_grad else None
            logits = _EngineFn.apply(core.anchor, logits, core, emb_edge)
        outputs = (logits,)
        mem_len = getattr(self.config, "mem_len", None)
        if mem_len is not None and mem_len > 0 and use_cache is True:          # xlnet.py:363-365, 406-407, 511-513: (logits, mems, ...)
            outputs = outputs + (_xl_new_mems(core, mems, mlen, B, K, mem_len),)
        if output_hidden_states:
            hs = tuple(h[:, mlen:] for h in core.hidden_states(B, K))
            outputs = outputs + (hs if gs is None else _xl_interleave(hs, gs),)
        if output_attentions:
            outputs = outputs + (tuple(a[:, :, mlen:] for a in core.xl_attentions(B, K, self.training)),)
        if labels is not None:                                        # xlnet.py:515-524
            if self.num_labels == 1:
                loss = torch.nn.functional.mse_loss(logits.view(-1), labels.to(logits.device).float().view(-1))
            else:
                loss = torch.nn.functional.cross_entropy(logits.view(-1, self.num_labels), labels.to(logits.device).view(-1))
            outputs = (loss,) + outputs
        return outputs
