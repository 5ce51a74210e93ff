"""TEST INFRASTRUCTURE -- pure-torch fp32 CPU restatement of the MAG-BERT hot path.

No ``transformers`` import and no reference file is needed: this module travels to the
GPU box as (a) the parity checker for the HIP path and (b) the timed CPU baseline
(``bench.py`` ``cpu_baseline`` leg, kind="port").  It is checked against the reference's
own Python (run through the import shim) by ``oracle/make_golden.py`` and pinned by
``tests/golden/*.npz``.

Each class cites the reference file:line (relative to /root/reference) or, for the
third-party layers the reference only imports, the call site in the reference plus the
transformers==3.0.2 arithmetic it executes (requirements.txt:348).

Module / parameter names reproduce the reference's state-dict keys exactly:
  bert.embeddings.{word,position,token_type}_embeddings.weight, bert.embeddings.LayerNorm.*
  bert.encoder.layer.{i}.attention.self.{query,key,value}.*, .attention.output.{dense,LayerNorm}.*
  bert.encoder.layer.{i}.intermediate.dense.*, .output.{dense,LayerNorm}.*
  bert.pooler.dense.*, bert.MAG.{W_hv,W_ha,W_v,W_a,LayerNorm}.*, classifier.*
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


class BertConfigLite(object):
    """The subset of transformers.BertConfig (bert-base-uncased defaults) the path reads."""

    def __init__(self, vocab_size=30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                 intermediate_size=3072, hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1,
                 max_position_embeddings=512, type_vocab_size=2, layer_norm_eps=1e-12, num_labels=1,
                 initializer_range=0.02):
        self.vocab_size = vocab_size
        self.hidden_size = hidden_size
        self.num_hidden_layers = num_hidden_layers
        self.num_attention_heads = num_attention_heads
        self.intermediate_size = intermediate_size
        self.hidden_dropout_prob = hidden_dropout_prob
        self.attention_probs_dropout_prob = attention_probs_dropout_prob
        self.max_position_embeddings = max_position_embeddings
        self.type_vocab_size = type_vocab_size
        self.layer_norm_eps = layer_norm_eps
        self.num_labels = num_labels
        self.initializer_range = initializer_range


class MultimodalConfig(object):
    """multimodal_driver.py:76-79."""

    def __init__(self, beta_shift, dropout_prob):
        self.beta_shift = beta_shift
        self.dropout_prob = dropout_prob


class MAG(nn.Module):
    """Multimodal Adaptation Gate -- modeling.py:6-51 restated.

    Dimensions are constructor arguments instead of module globals
    (global_configs.py:9-17); everything else follows modeling.py line by line.
    """

    def __init__(self, hidden_size, beta_shift, dropout_prob, visual_dim=47, acoustic_dim=74):
        super().__init__()
        self.W_hv = nn.Linear(visual_dim + hidden_size, hidden_size)      # modeling.py:15
        self.W_ha = nn.Linear(acoustic_dim + hidden_size, hidden_size)    # modeling.py:16
        self.W_v = nn.Linear(visual_dim, hidden_size)                     # modeling.py:18
        self.W_a = nn.Linear(acoustic_dim, hidden_size)                   # modeling.py:19
        self.beta_shift = beta_shift
        self.LayerNorm = nn.LayerNorm(hidden_size)                        # modeling.py:22 (eps 1e-5 default)
        self.dropout = nn.Dropout(dropout_prob)                           # modeling.py:23

    def forward(self, text_embedding, visual, acoustic):
        eps = 1e-6                                                                            # modeling.py:26
        weight_v = F.relu(self.W_hv(torch.cat((visual, text_embedding), dim=-1)))           # :27
        weight_a = F.relu(self.W_ha(torch.cat((acoustic, text_embedding), dim=-1)))         # :28
        h_m = weight_v * self.W_v(visual) + weight_a * self.W_a(acoustic)                    # :30
        em_norm = text_embedding.norm(2, dim=-1)                                              # :32
        hm_norm = h_m.norm(2, dim=-1)                                                         # :33
        hm_norm = torch.where(hm_norm == 0, torch.ones_like(hm_norm), hm_norm)                # :35-36
        thresh_hold = (em_norm / (hm_norm + eps)) * self.beta_shift                           # :38
        alpha = torch.min(thresh_hold, torch.ones_like(thresh_hold)).unsqueeze(dim=-1)        # :40-43
        acoustic_vis_embedding = alpha * h_m                                                  # :45
        return self.dropout(self.LayerNorm(acoustic_vis_embedding + text_embedding))          # :47-49


class BertEmbeddings(nn.Module):
    """transformers==3.0.2 BertEmbeddings as called at bert.py:81,211-216:
    LN_{eps}(word[ids] + position[0..L) + token_type[seg]) -> dropout."""

    def __init__(self, c):
        super().__init__()
        # 3.0.2: nn.Embedding(vocab, hidden, padding_idx=config.pad_token_id): the [PAD] row (id 0) never receives a gradient
        self.word_embeddings = nn.Embedding(c.vocab_size, c.hidden_size, padding_idx=getattr(c, "pad_token_id", 0))
        self.position_embeddings = nn.Embedding(c.max_position_embeddings, c.hidden_size)
        self.token_type_embeddings = nn.Embedding(c.type_vocab_size, c.hidden_size)
        self.LayerNorm = nn.LayerNorm(c.hidden_size, eps=c.layer_norm_eps)
        self.dropout = nn.Dropout(c.hidden_dropout_prob)

    def forward(self, input_ids, token_type_ids, inputs_embeds=None, position_ids=None):
        # inputs_embeds (bert.py:211-216 passes it through): used in place of word_embeddings(input_ids)
        # position_ids (same call): rows of the position table; 3.0.2 default = arange(seq_len) for every sample
        L = token_type_ids.shape[1]
        pos = torch.arange(L, dtype=torch.long, device=token_type_ids.device).unsqueeze(0).expand_as(token_type_ids)
        if position_ids is not None:
            pos = position_ids.to(torch.long).expand_as(token_type_ids)
        words = self.word_embeddings(input_ids) if inputs_embeds is None else inputs_embeds
        e = words + self.position_embeddings(pos) + self.token_type_embeddings(token_type_ids)
        return self.dropout(self.LayerNorm(e))


class BertSelfAttention(nn.Module):
    """3.0.2 BertSelfAttention (call: bert.py:221-229): softmax(QK^T/sqrt(dh)+mask) -> dropout -> .V"""

    def __init__(self, c):
        super().__init__()
        self.nh = c.num_attention_heads
        self.dh = c.hidden_size // c.num_attention_heads
        self.query = nn.Linear(c.hidden_size, c.hidden_size)
        self.key = nn.Linear(c.hidden_size, c.hidden_size)
        self.value = nn.Linear(c.hidden_size, c.hidden_size)
        self.dropout = nn.Dropout(c.attention_probs_dropout_prob)

    def _split(self, x):
        B, L, _ = x.shape
        return x.view(B, L, self.nh, self.dh).permute(0, 2, 1, 3)

    def forward(self, x, ext_mask, head_mask=None):
        q, k, v = self._split(self.query(x)), self._split(self.key(x)), self._split(self.value(x))
        s = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(self.dh)
        s = s + ext_mask
        p = self.dropout(F.softmax(s, dim=-1))
        if head_mask is not None:          # 3.0.2: attention_probs = attention_probs * head_mask, after the dropout
            p = p * head_mask
        self.last_probs = p                # what output_attentions returns
        ctx = torch.matmul(p, v).permute(0, 2, 1, 3).contiguous()
        return ctx.view(x.shape[0], x.shape[1], self.nh * self.dh)


class BertSelfOutput(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.dense = nn.Linear(c.hidden_size, c.hidden_size)
        self.LayerNorm = nn.LayerNorm(c.hidden_size, eps=c.layer_norm_eps)
        self.dropout = nn.Dropout(c.hidden_dropout_prob)

    def forward(self, h, inp):
        return self.LayerNorm(self.dropout(self.dense(h)) + inp)


class BertAttention(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.self = BertSelfAttention(c)
        self.output = BertSelfOutput(c)

    def forward(self, x, ext_mask, head_mask=None):
        return self.output(self.self(x, ext_mask, head_mask), x)


class BertIntermediate(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.dense = nn.Linear(c.hidden_size, c.intermediate_size)

    def forward(self, x):
        return F.gelu(self.dense(x))          # erf GELU (3.0.2 ACT2FN["gelu"])


class BertOutput(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.dense = nn.Linear(c.intermediate_size, c.hidden_size)
        self.LayerNorm = nn.LayerNorm(c.hidden_size, eps=c.layer_norm_eps)
        self.dropout = nn.Dropout(c.hidden_dropout_prob)

    def forward(self, h, inp):
        return self.LayerNorm(self.dropout(self.dense(h)) + inp)


class BertLayer(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.attention = BertAttention(c)
        self.intermediate = BertIntermediate(c)
        self.output = BertOutput(c)

    def forward(self, x, ext_mask, head_mask=None):
        a = self.attention(x, ext_mask, head_mask)
        return self.output(self.intermediate(a), a)


class BertEncoder(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.layer = nn.ModuleList([BertLayer(c) for _ in range(c.num_hidden_layers)])

    def forward(self, x, ext_mask, head_mask=None):
        for i, lyr in enumerate(self.layer):
            x = lyr(x, ext_mask, None if head_mask is None else head_mask[i])
        return x


class BertPooler(nn.Module):
    """3.0.2 BertPooler (bert.py:83,231): tanh(dense(h[:,0]))."""

    def __init__(self, c):
        super().__init__()
        self.dense = nn.Linear(c.hidden_size, c.hidden_size)

    def forward(self, h):
        return torch.tanh(self.dense(h[:, 0]))


class MAG_BertModel(nn.Module):
    """bert.py:76-237: embeddings -> MAG -> encoder -> pooler."""

    def __init__(self, config, multimodal_config, visual_dim=47, acoustic_dim=74):
        super().__init__()
        self.config = config
        self.embeddings = BertEmbeddings(config)                 # bert.py:81
        self.encoder = BertEncoder(config)                       # bert.py:82
        self.pooler = BertPooler(config)                         # bert.py:83
        self.MAG = MAG(config.hidden_size, multimodal_config.beta_shift, multimodal_config.dropout_prob,
                       visual_dim, acoustic_dim)                 # bert.py:84-88

    def forward(self, input_ids, visual, acoustic, attention_mask=None, token_type_ids=None, head_mask=None,
                inputs_embeds=None, position_ids=None):
        shape = input_ids.shape if input_ids is not None else inputs_embeds.shape[:-1]      # bert.py:158-168
        dev = visual.device
        if attention_mask is None:
            attention_mask = torch.ones(shape, dtype=torch.long, device=dev)     # bert.py:173-174
        if token_type_ids is None:
            token_type_ids = torch.zeros(shape, dtype=torch.long, device=dev)    # bert.py:175-177
        if head_mask is not None:
            # 3.0.2 get_head_mask (bert.py:206-207): [nh] -> every layer, [NL][nh] as is; broadcast over [B][nh][L][L]
            head_mask = head_mask.to(torch.float32)
            if head_mask.dim() == 1:
                head_mask = head_mask[None].expand(self.config.num_hidden_layers, -1)
            head_mask = head_mask[:, None, :, None, None]
        # 3.0.2 get_extended_attention_mask (bert.py:180-182): (1 - mask)[:,None,None,:] * -10000.0
        ext = (1.0 - attention_mask[:, None, None, :].to(torch.float32)) * -10000.0
        emb = self.embeddings(input_ids, token_type_ids, inputs_embeds, position_ids)     # bert.py:211-216
        fused = self.MAG(emb, visual, acoustic)                  # bert.py:219
        seq = self.encoder(fused, ext, head_mask)                # bert.py:221-229
        pooled = self.pooler(seq)                                # bert.py:231
        return seq, pooled


class MAG_BertForSequenceClassification(nn.Module):
    """bert.py:240-324."""

    def __init__(self, config, multimodal_config, visual_dim=47, acoustic_dim=74):
        super().__init__()
        self.num_labels = config.num_labels
        self.bert = MAG_BertModel(config, multimodal_config, visual_dim, acoustic_dim)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)               # bert.py:246
        self.classifier = nn.Linear(config.hidden_size, config.num_labels)  # bert.py:247

    def forward(self, input_ids, visual, acoustic, attention_mask=None, token_type_ids=None, labels=None, head_mask=None,
                inputs_embeds=None, position_ids=None):
        seq, pooled = self.bert(input_ids, visual, acoustic, attention_mask, token_type_ids, head_mask, inputs_embeds, position_ids)
        logits = self.classifier(self.dropout(pooled))                      # bert.py:304-307
        outputs = (logits,)
        if labels is not None:                                              # bert.py:313-322
            if self.num_labels == 1:
                loss = F.mse_loss(logits.view(-1), labels.view(-1))
            else:
                loss = F.cross_entropy(logits.view(-1, self.num_labels), labels.view(-1))
            outputs = (loss,) + outputs
        return outputs


def load_deterministic(model, mode="test"):
    """Fill every parameter from oracle.weights.make_param keyed by its state-dict name."""
    from . import weights
    with torch.no_grad():
        for name, p in model.named_parameters():
            p.copy_(torch.from_numpy(weights.make_param(name, tuple(p.shape), mode)))
    return model


def set_dropout(model, p_hidden=None, p_attn=None, p_mag=None):
    """Override dropout probabilities in place (parity runs use p=0 in train mode)."""
    for name, m in model.named_modules():
        if isinstance(m, nn.Dropout):
            if name.endswith("MAG.dropout"):
                if p_mag is not None:
                    m.p = p_mag
            elif name.endswith("attention.self.dropout"):
                if p_attn is not None:
                    m.p = p_attn
            elif p_hidden is not None:
                m.p = p_hidden
    return model
