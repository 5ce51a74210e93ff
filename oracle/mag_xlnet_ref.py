"""TEST INFRASTRUCTURE -- pure-torch fp32 CPU restatement of the MAG-XLNet hot path (BASELINE.json config 4).

Restates MAG_XLNetModel.forward / MAG_XLNetForSequenceClassification.forward (xlnet.py:148-429, 443-527) for the only
configuration the driver exercises (xlnet-base-cased: attn_type "bi", bi_data False, clamp_len -1, target_mapping / the query stream since round 6;
multimodal_driver.py:363-370; round 5: mems / mem_len, xlnet.py:81-91, 244-245, 276-293, 317-323, 363-365) and the transformers==3.0.2 XLNetLayer
(XLNetRelativeAttention + XLNetFeedForward) and SequenceSummary it calls (xlnet.py:30,374-385,438,508).  Checked against
the reference's own Python by oracle/make_golden.py (G6 fixtures).  Works in the reference's [L, B, .] layout internally.

State-dict keys = the reference's: transformer.word_embedding.weight, transformer.mask_emb,
transformer.layer.{i}.rel_attn.{q,k,v,o,r,r_r_bias,r_s_bias,r_w_bias,seg_embed,layer_norm.*},
transformer.layer.{i}.ff.{layer_norm,layer_1,layer_2}.*, transformer.MAG.*, sequence_summary.summary.*, logits_proj.*
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .mag_bert_ref import MAG, MultimodalConfig  # noqa: F401


class XLNetConfigLite(object):
    def __init__(self, vocab_size=32000, d_model=768, n_layer=12, n_head=12, d_inner=3072, dropout=0.1,
                 layer_norm_eps=1e-12, summary_last_dropout=0.1, num_labels=1, initializer_range=0.02):
        self.vocab_size = vocab_size
        self.d_model = d_model
        self.hidden_size = d_model
        self.n_layer = n_layer
        self.n_head = n_head
        self.d_head = d_model // n_head
        self.d_inner = d_inner
        self.dropout = dropout
        self.layer_norm_eps = layer_norm_eps
        self.summary_last_dropout = summary_last_dropout
        self.num_labels = num_labels
        self.initializer_range = initializer_range


class XLNetRelativeAttention(nn.Module):
    """3.0.2 XLNetRelativeAttention: the content stream, and the query stream when `g` is given (two-stream attention)."""

    def __init__(self, c):
        super().__init__()
        self.n_head, self.d_head, self.scale = c.n_head, c.d_head, 1 / (c.d_head ** 0.5)
        for n in ("q", "k", "v", "o", "r"):
            setattr(self, n, nn.Parameter(torch.zeros(c.d_model, c.n_head, c.d_head)))
        self.r_r_bias = nn.Parameter(torch.zeros(c.n_head, c.d_head))
        self.r_s_bias = nn.Parameter(torch.zeros(c.n_head, c.d_head))
        self.r_w_bias = nn.Parameter(torch.zeros(c.n_head, c.d_head))
        self.seg_embed = nn.Parameter(torch.zeros(2, c.n_head, c.d_head))
        self.layer_norm = nn.LayerNorm(c.d_model, eps=c.layer_norm_eps)
        self.dropout = nn.Dropout(c.dropout)

    @staticmethod
    def rel_shift_bnij(x, klen):
        s = x.shape
        x = x.reshape(s[0], s[1], s[3], s[2])[:, :, 1:, :]
        x = x.reshape(s[0], s[1], s[2], s[3] - 1)
        return x[:, :, :, :klen]          # => bd[i, j] = raw[i, L - i + j]

    def rel_attn_core(self, q, k, v, kr, seg_mat, attn_mask, head_mask):
        """3.0.2 XLNetRelativeAttention.rel_attn_core: content + position + segment scores, mask, softmax, dropout, P.V"""
        ac = torch.einsum("ibnd,jbnd->bnij", q + self.r_w_bias, k)
        bd = self.rel_shift_bnij(torch.einsum("ibnd,jbnd->bnij", q + self.r_r_bias, kr), ac.shape[3])
        ef = torch.einsum("ibnd,snd->ibns", q + self.r_s_bias, self.seg_embed)
        ef = torch.einsum("ijbs,ibns->bnij", seg_mat, ef)
        score = (ac + bd + ef) * self.scale
        score = score - 1e30 * torch.einsum("ijbn->bnij", attn_mask)
        p = self.dropout(F.softmax(score, dim=3))
        if head_mask is not None:           # xlnet.py:383 -> rel_attn_core: attn_prob = attn_prob * head_mask, after the dropout
            p = p * head_mask.view(1, -1, 1, 1)
        return torch.einsum("bnij,jbnd->ibnd", p, v), p

    def forward(self, h, attn_mask, r, seg_mat, head_mask=None, mems=None, g=None, attn_mask_g=None, target_mapping=None):
        # 3.0.2 XLNetRelativeAttention.forward: keys and values over cat([mems, h]) (the cached hidden states of the previous
        # segment, xlnet.py:374-385 passes mems[i]); queries over h only.  g (the query stream, xlnet.py:306-313, 374-385 with
        # target_mapping [M, L, B]): queries from g, mapped onto the L positions, attend to the CONTENT stream's keys / values under
        # attn_mask_g (no self-exemption), mapped back to the M prediction rows; same o projection and LayerNorm
        cat = h if mems is None else torch.cat([mems, h], dim=0)
        q = torch.einsum("ibh,hnd->ibnd", h, self.q)
        k = torch.einsum("ibh,hnd->ibnd", cat, self.k)
        v = torch.einsum("ibh,hnd->ibnd", cat, self.v)
        kr = torch.einsum("ibh,hnd->ibnd", r, self.r)
        vec, p = self.rel_attn_core(q, k, v, kr, seg_mat, attn_mask, head_mask)
        self.last_probs = p                 # what output_attentions returns (xlnet.py:387-427): [B, n_head, L, L], after the dropout
        out = self.dropout(torch.einsum("ibnd,hnd->ibh", vec, self.o))
        out_h = self.layer_norm(out + h)
        if g is None:
            return out_h
        qg = torch.einsum("ibh,hnd->ibnd", g, self.q)
        qg = torch.einsum("mbnd,mlb->lbnd", qg, target_mapping)
        vec_g, _ = self.rel_attn_core(qg, k, v, kr, seg_mat, attn_mask_g, head_mask)
        vec_g = torch.einsum("lbnd,mlb->mbnd", vec_g, target_mapping)
        out_g = self.dropout(torch.einsum("ibnd,hnd->ibh", vec_g, self.o))
        return out_h, self.layer_norm(out_g + g)


class XLNetFeedForward(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.layer_norm = nn.LayerNorm(c.d_model, eps=c.layer_norm_eps)
        self.layer_1 = nn.Linear(c.d_model, c.d_inner)
        self.layer_2 = nn.Linear(c.d_inner, c.d_model)
        self.dropout = nn.Dropout(c.dropout)

    def forward(self, inp):
        out = self.dropout(F.gelu(self.layer_1(inp)))
        out = self.dropout(self.layer_2(out))
        return self.layer_norm(out + inp)


class XLNetLayer(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.rel_attn = XLNetRelativeAttention(c)
        self.ff = XLNetFeedForward(c)

    def forward(self, h, attn_mask, r, seg_mat, head_mask=None, mems=None, g=None, attn_mask_g=None, target_mapping=None):
        if g is None:
            return self.ff(self.rel_attn(h, attn_mask, r, seg_mat, head_mask, mems))
        oh, og = self.rel_attn(h, attn_mask, r, seg_mat, head_mask, mems, g, attn_mask_g, target_mapping)
        return self.ff(oh), self.ff(og)                                                 # 3.0.2 XLNetLayer: the same feed-forward on both streams


class MAG_XLNetModel(nn.Module):
    """xlnet.py:15-429 (driver configuration only)."""

    def __init__(self, config, multimodal_config, visual_dim=47, acoustic_dim=74, injection_index=1):
        super().__init__()
        self.d_model, self.n_layer, self.injection_index = config.d_model, config.n_layer, injection_index
        self.word_embedding = nn.Embedding(config.vocab_size, config.d_model)          # xlnet.py:28
        self.mask_emb = nn.Parameter(torch.zeros(1, 1, config.d_model))                # xlnet.py:29 (unused here)
        self.layer = nn.ModuleList([XLNetLayer(config) for _ in range(config.n_layer)])
        self.dropout = nn.Dropout(config.dropout)
        self.MAG = MAG(config.d_model, multimodal_config.beta_shift, multimodal_config.dropout_prob, visual_dim, acoustic_dim)

    def relative_positional_encoding(self, qlen, klen, bsz):
        """xlnet.py:104-146 with attn_type "bi", bi_data False, clamp_len -1; positional_embedding :93-102."""
        freq_seq = torch.arange(0, self.d_model, 2.0, dtype=torch.float)
        inv_freq = 1 / torch.pow(10000, (freq_seq / self.d_model))
        pos_seq = torch.arange(klen, -qlen, -1.0)
        sinusoid = torch.einsum("i,d->id", pos_seq, inv_freq)
        pos_emb = torch.cat([torch.sin(sinusoid), torch.cos(sinusoid)], dim=-1)
        return pos_emb[:, None, :].expand(-1, bsz, -1)

    @staticmethod
    def cache_mem(curr_out, prev_mem, mem_len):
        """xlnet.py:81-91 (reuse_len None): the last mem_len rows of cat([prev_mem, curr_out]), detached"""
        new_mem = curr_out[-mem_len:] if prev_mem is None else torch.cat([prev_mem, curr_out], dim=0)[-mem_len:]
        return new_mem.detach()

    def forward(self, input_ids, visual, acoustic, attention_mask, token_type_ids, head_mask=None, inputs_embeds=None,
                perm_mask=None, input_mask=None, mems=None, mem_len=None, target_mapping=None):
        """target_mapping: [B, M, L] (xlnet.py:238-240 permutes it to [M, L, B]) -> the query stream g runs next to h and the model
        returns ITS last state, [B, M, d] (xlnet.py:306-313, 374-399);
        mems: list of n_layer tensors [mlen, B, d] (the hidden states cached from the previous segment, xlnet.py:244-245,
        374-385) or None; mem_len: > 0 -> self.new_mems is set to the n_layer tensors cache_mem produces (xlnet.py:363-365,
        use_cache True)"""
        if head_mask is not None:            # xlnet.py:340-353: [n_head] -> every layer, [n_layer][n_head] as is
            head_mask = head_mask.to(torch.float32)
            if head_mask.dim() == 1:
                head_mask = head_mask[None].expand(self.n_layer, -1)
        if inputs_embeds is not None:        # xlnet.py:208-210: [B, L, d] -> [L, B, d]; used instead of the table (xlnet.py:301-305)
            emb = inputs_embeds.transpose(0, 1).contiguous()
            L, B = emb.shape[0], emb.shape[1]
        else:
            ids = input_ids.transpose(0, 1).contiguous()                                # xlnet.py:206
            L, B = ids.shape
        visual = visual.transpose(0, 1).contiguous()                                    # xlnet.py:215-216
        acoustic = acoustic.transpose(0, 1).contiguous()
        seg = token_type_ids.transpose(0, 1).contiguous()
        assert input_mask is None or attention_mask is None                            # xlnet.py:258-262
        input_mask = input_mask.transpose(0, 1).contiguous().float() if input_mask is not None else None      # xlnet.py:236
        perm_mask = perm_mask.permute(1, 2, 0).contiguous().float() if perm_mask is not None else None        # xlnet.py:237: [i, j, b]
        mlen = mems[0].shape[0] if mems is not None and mems[0] is not None else 0      # xlnet.py:244-245; klen = mlen + L
        if input_mask is None and attention_mask is not None:
            input_mask = 1.0 - attention_mask.transpose(0, 1).contiguous().float()      # xlnet.py:263-264
        if input_mask is not None and perm_mask is not None:                            # xlnet.py:265-272
            data_mask = input_mask[None] + perm_mask
        elif input_mask is not None:
            data_mask = input_mask[None]
        elif perm_mask is not None:
            data_mask = perm_mask
        else:
            data_mask = None
        if data_mask is not None:
            if mlen > 0:                                                                # xlnet.py:276-280: all mems can be attended to
                data_mask = torch.cat([torch.zeros(data_mask.shape[0], mlen, B), data_mask], dim=1)
            attn_mask = (data_mask[:, :, :, None] > 0).float()                          # xlnet.py:274-286 : [1 | L, klen, B, 1]
            eye = torch.eye(L) if mlen == 0 else torch.cat([torch.zeros(L, mlen), torch.eye(L)], dim=-1)      # xlnet.py:289-293
            non_tgt = ((attn_mask - eye[:, :, None, None]) > 0).float()                 # xlnet.py:288-296 : [L, klen, B, 1]
            attn_mask_g = attn_mask
        else:
            non_tgt = torch.zeros(L, L + mlen, B, 1)                                    # (attn_mask None: nothing is masked)
            attn_mask_g = torch.zeros(1, L + mlen, B, 1)
        h = self.dropout(emb if inputs_embeds is not None else self.word_embedding(ids))        # xlnet.py:301-305
        cat_ids = seg if mlen == 0 else torch.cat([torch.zeros(mlen, B, dtype=seg.dtype), seg], dim=0)      # xlnet.py:317-323: mem_pad
        seg_mat = (seg[:, None] != cat_ids[None, :]).long()                             # xlnet.py:326
        seg_mat = F.one_hot(seg_mat, num_classes=2).float()                             # xlnet.py:327
        dt = self.word_embedding.weight.dtype          # float32; float64 when a conditioning analysis runs the oracle in double
        non_tgt, seg_mat = non_tgt.to(dt), seg_mat.to(dt)
        g = tm = None
        if target_mapping is not None:                                                  # xlnet.py:238-240, 306-313
            tm = target_mapping.permute(1, 2, 0).contiguous().to(dt)
            g = self.dropout(self.mask_emb.expand(tm.shape[0], B, -1).to(dt))
            attn_mask_g = attn_mask_g.to(dt).expand(L, -1, -1, -1) if attn_mask_g.shape[0] == 1 else attn_mask_g.to(dt)
        self.hidden_g = []
        pos_emb = self.dropout(self.relative_positional_encoding(L, L + mlen, B).to(dt))       # xlnet.py:332-333
        self.new_mems = None if not mem_len else []
        for i, layer in enumerate(self.layer):
            if mem_len:                                                                 # xlnet.py:363-365
                self.new_mems.append(self.cache_mem(h, None if mems is None else mems[i], mem_len))
            if i == self.injection_index:
                h = self.MAG(h, visual, acoustic)                                       # xlnet.py:371-372
            if g is None:
                h = layer(h, non_tgt, pos_emb, seg_mat, None if head_mask is None else head_mask[i],
                          None if mems is None else mems[i])                            # xlnet.py:374-385
            else:
                self.hidden_g.append(g)
                h, g = layer(h, non_tgt, pos_emb, seg_mat, None if head_mask is None else head_mask[i],
                             None if mems is None else mems[i], g, attn_mask_g, tm)
        if g is not None:
            self.hidden_g.append(g)
        return self.dropout(g if g is not None else h).permute(1, 0, 2).contiguous()   # xlnet.py:396-399


class SequenceSummary(nn.Module):
    """3.0.2 SequenceSummary for xlnet-base-cased: summary_type "last", use_proj, tanh, last dropout 0.1."""

    def __init__(self, c):
        super().__init__()
        self.summary = nn.Linear(c.d_model, c.d_model)
        self.last_dropout = nn.Dropout(c.summary_last_dropout)

    def forward(self, hidden):
        return self.last_dropout(torch.tanh(self.summary(hidden[:, -1])))


class MAG_XLNetForSequenceClassification(nn.Module):
    """xlnet.py:432-527."""

    def __init__(self, config, multimodal_config, visual_dim=47, acoustic_dim=74, injection_index=1):
        super().__init__()
        self.num_labels = config.num_labels
        self.transformer = MAG_XLNetModel(config, multimodal_config, visual_dim, acoustic_dim, injection_index)
        self.sequence_summary = SequenceSummary(config)
        self.logits_proj = nn.Linear(config.d_model, config.num_labels)

    def forward(self, input_ids, visual, acoustic, attention_mask, token_type_ids, labels=None, head_mask=None, inputs_embeds=None,
                perm_mask=None, input_mask=None, mems=None, mem_len=None, target_mapping=None):
        out = self.transformer(input_ids, visual, acoustic, attention_mask, token_type_ids, head_mask, inputs_embeds, perm_mask, input_mask,
                               mems, mem_len, target_mapping)
        logits = self.logits_proj(self.sequence_summary(out))                           # xlnet.py:506-509
        outputs = (logits,)
        if labels is not None:                                                          # xlnet.py:515-524
            if self.num_labels == 1:
                outputs = (F.mse_loss(logits.view(-1), labels.view(-1)),) + outputs
            else:
                outputs = (F.cross_entropy(logits.view(-1, self.num_labels), labels.view(-1)),) + outputs
        return outputs


def load_deterministic(model, mode="test"):
    from . import weights
    with torch.no_grad():
        for name, p in model.named_parameters():
            p.copy_(torch.from_numpy(weights.make_param(name, tuple(p.shape), mode)))
    return model


def set_dropout(model, p_hidden=None, p_mag=None):
    for name, m in model.named_modules():
        if isinstance(m, nn.Dropout):
            if name.endswith("MAG.dropout"):
                if p_mag is not None:
                    m.p = p_mag
            elif p_hidden is not None:
                m.p = p_hidden
    return model
