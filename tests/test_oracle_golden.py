"""CPU: the oracle (oracle/mag_bert_ref.py) against the golden vectors produced by the reference's own Python
(tests/golden/*.npz, generator: oracle/make_golden.py)."""
import numpy as np
import torch

from oracle import mag_bert_ref as R
from oracle import weights


def _mag(V, mode, beta):
    m = R.MAG(768, beta, 0.5, V, 74)
    with torch.no_grad():
        for n, p in m.named_parameters():
            p.copy_(torch.from_numpy(weights.make_param("bert.MAG." + n, tuple(p.shape), mode)))
    return m.eval()


def test_mag_forward_backward_matches_reference(golden):
    g = golden["g1_mag"]
    for V in (47, 35):
        for mode in ("test", "init"):
            for beta in (1.0, 1e-3):
                key = "V%d_%s_b%g" % (V, mode, beta)
                m = _mag(V, mode, beta)
                e = torch.tensor(weights.uniform("mag.e", (2, 8, 768), -1.5, 1.5), requires_grad=True)
                b = weights.synthetic_bert_batch(2, 8, V, 74, seed=77, min_len=2)
                v = torch.tensor(b["visual"], requires_grad=True)
                a = torch.tensor(b["acoustic"], requires_grad=True)
                y = m(e, v, a)
                (y * torch.from_numpy(weights.uniform("mag.dy", tuple(y.shape)))).sum().backward()
                np.testing.assert_allclose(y.detach().numpy(), g[key + "/out"], atol=1e-5)
                np.testing.assert_allclose(e.grad.numpy(), g[key + "/d_text"], atol=1e-5)
                np.testing.assert_allclose(v.grad.numpy(), g[key + "/d_visual"], atol=1e-5)
                np.testing.assert_allclose(a.grad.numpy(), g[key + "/d_acoustic"], atol=1e-5)
                for n, p in m.named_parameters():
                    np.testing.assert_allclose(float(p.grad.norm()), float(g[key + "/gnorm/" + n]), rtol=1e-4, atol=1e-6)
                    np.testing.assert_allclose(weights.strided_sample(p.grad.numpy()), g[key + "/gslice/" + n], atol=1e-5)
                assert torch.isfinite(e.grad).all() and torch.isfinite(v.grad).all()


def test_mag_zero_rows_take_the_hm_zero_branch():
    # init-law biases (0) + zero modality rows -> h_m == 0 exactly -> hm_norm replaced by 1 (modeling.py:35-36)
    m = _mag(47, "init", 1.0)
    b = weights.synthetic_bert_batch(2, 8, 47, 74, seed=77, min_len=2)
    v, a = torch.tensor(b["visual"]), torch.tensor(b["acoustic"])
    h = torch.relu(m.W_hv(torch.cat((v, torch.zeros(2, 8, 768)), -1))) * m.W_v(v)
    assert float(h[0, 0].abs().max()) == 0.0


def _model(V, layers=12, mode="test", p_mag=0.5):
    m = R.MAG_BertForSequenceClassification(R.BertConfigLite(num_hidden_layers=layers), R.MultimodalConfig(1.0, p_mag), V, 74)
    return R.load_deterministic(m, mode)


def test_embeddings_and_layer(golden):
    g = golden["g2g3_embeddings_layer"]
    m = _model(47, layers=1).eval()
    b = weights.synthetic_bert_batch(2, 8, 47, 74, seed=5, min_len=2)
    seg = torch.from_numpy(g["emb/segment_ids"])
    e = m.bert.embeddings(torch.from_numpy(b["input_ids"]), seg)
    np.testing.assert_allclose(e.detach().numpy(), g["emb/out"], atol=1e-5)
    x = torch.tensor(weights.uniform("layer.x", (2, 8, 768), -1.0, 1.0), requires_grad=True)
    ext = (1.0 - torch.from_numpy(b["input_mask"])[:, None, None, :].float()) * -10000.0
    lyr = m.bert.encoder.layer[0]
    y = lyr(x, ext)
    (y * torch.from_numpy(weights.uniform("layer.dy", tuple(y.shape)))).sum().backward()
    np.testing.assert_allclose(y.detach().numpy(), g["layer/out"], atol=2e-5)
    np.testing.assert_allclose(x.grad.numpy(), g["layer/dx"], atol=2e-5)
    for n, p in lyr.named_parameters():
        np.testing.assert_allclose(float(p.grad.norm()), float(g["layer/gnorm/" + n]), rtol=1e-4, atol=1e-6)


def test_full_model_logits_and_grads(golden):
    g = golden["g4g5_full_model"]
    torch.set_num_threads(8)
    m = _model(47).eval()
    b = weights.synthetic_bert_batch(4, 50, 47, 74, seed=11)
    with torch.no_grad():
        logits = m(torch.from_numpy(b["input_ids"]), torch.from_numpy(b["visual"]), torch.from_numpy(b["acoustic"]),
                   torch.from_numpy(b["input_mask"]), torch.from_numpy(b["segment_ids"]))[0]
    np.testing.assert_allclose(logits.numpy(), g["logits/B4_L50_V47_seed11"], atol=2e-5)
    # train mode, every dropout p = 0
    m = R.set_dropout(_model(47, p_mag=0.0), 0.0, 0.0, 0.0).train()
    b = weights.synthetic_bert_batch(4, 50, 47, 74, seed=21)
    logits = m(torch.from_numpy(b["input_ids"]), torch.from_numpy(b["visual"]), torch.from_numpy(b["acoustic"]),
               torch.from_numpy(b["input_mask"]), torch.from_numpy(b["segment_ids"]))[0]
    loss = torch.nn.functional.mse_loss(logits.view(-1), torch.from_numpy(b["label_ids"]).view(-1))
    loss.backward()
    assert abs(float(loss) - float(g["train/loss_B4_L50_seed21"])) < 1e-5
    for n, p in m.named_parameters():
        np.testing.assert_allclose(float(p.grad.norm()), float(g["train/gnorm/" + n]), rtol=2e-3, atol=1e-6)


def test_xlnet_oracle_logits_and_grads(golden):
    """G6: the MAG-XLNet restatement vs outputs of the reference's own xlnet.py (oracle/make_golden.py gen_xlnet)."""
    from oracle import mag_xlnet_ref as X
    g = golden["g6_xlnet"]
    torch.set_num_threads(8)
    t = lambda b: (torch.from_numpy(b["input_ids"]), torch.from_numpy(b["visual"]), torch.from_numpy(b["acoustic"]),
                   torch.from_numpy(b["input_mask"]), torch.from_numpy(b["segment_ids"]))
    m = X.load_deterministic(X.MAG_XLNetForSequenceClassification(X.XLNetConfigLite(), X.MultimodalConfig(1.0, 0.5), 47, 74)).eval()
    with torch.no_grad():
        logits = m(*t(weights.synthetic_xlnet_batch(4, 50, 47, 74, seed=31)))[0]
    np.testing.assert_allclose(logits.numpy(), g["logits/B4_L50_seed31"], atol=2e-5)
    m = X.set_dropout(m, 0.0, 0.0).train()
    b = weights.synthetic_xlnet_batch(4, 50, 47, 74, seed=33)
    loss = torch.nn.functional.mse_loss(m(*t(b))[0].view(-1), torch.from_numpy(b["label_ids"]).view(-1))
    loss.backward()
    assert abs(float(loss) - float(g["train/loss_B4_L50_seed33"])) < 1e-5
    assert m.transformer.mask_emb.grad is None                      # unused parameter (xlnet.py:29): frozen in the engine too
    for n, p in m.named_parameters():
        if p.grad is not None:
            np.testing.assert_allclose(float(p.grad.norm()), float(g["train/gnorm/" + n]), rtol=2e-3, atol=1e-6)
            np.testing.assert_allclose(weights.strided_sample(p.grad.numpy(), 16), g["train/gslice/" + n], rtol=5e-3, atol=1e-6)


def test_xlnet_oracle_mems(golden):
    """G6, round 5: the oracle's mems / mem_len path (xlnet.py:81-91, 244-245, 276-293, 317-323, 363-385) vs the reference's own outputs:
    segment 1 caches, segment 2 consumes (klen = mlen + L) and caches again -- logits of both segments, samples of the memories."""
    from oracle import mag_xlnet_ref as X
    g = golden["g6_xlnet"]
    torch.set_num_threads(8)
    B, L, ml, seed = 4, 24, 24, 36
    tag = "B%d_L%d_M%d_seed%d" % (B, L, ml, seed)
    t = lambda b: (torch.from_numpy(b["input_ids"]), torch.from_numpy(b["visual"]), torch.from_numpy(b["acoustic"]),
                   torch.from_numpy(b["input_mask"]), torch.from_numpy(b["segment_ids"]))
    m = X.load_deterministic(X.MAG_XLNetForSequenceClassification(X.XLNetConfigLite(), X.MultimodalConfig(1.0, 0.5), 47, 74)).eval()
    with torch.no_grad():
        l1 = m(*t(weights.synthetic_xlnet_batch(B, L, 47, 74, seed=seed)), mem_len=ml)[0]
        mems1 = m.transformer.new_mems
        l2 = m(*t(weights.synthetic_xlnet_batch(B, L, 47, 74, seed=seed + 100)), mems=mems1, mem_len=ml)[0]
        mems2 = m.transformer.new_mems
    np.testing.assert_allclose(l1.numpy(), g["mems/logits_seg1/" + tag], atol=2e-5)
    np.testing.assert_allclose(l2.numpy(), g["mems/logits_seg2/" + tag], atol=2e-5)
    assert len(mems2) == 12 and tuple(mems2[0].shape) == (ml, B, 768)
    for i in (0, 1, 2, 11):
        np.testing.assert_allclose(weights.strided_sample(mems1[i].numpy(), 64), g["mems/new_mems_seg1/%s/layer%d" % (tag, i)], atol=2e-5)
        np.testing.assert_allclose(weights.strided_sample(mems2[i].numpy(), 64), g["mems/new_mems_seg2/%s/layer%d" % (tag, i)], atol=2e-5)


def test_xlnet_oracle_query_stream(golden):
    """G6, round 6: target_mapping -> the two-stream attention (xlnet.py:238-240, 306-313, 374-399): the oracle's query stream output
    [B, M, 768] and the head's logits on it vs the reference's own outputs."""
    from oracle import mag_xlnet_ref as X
    g = golden["g6_xlnet"]
    torch.set_num_threads(8)
    B, L, M, seed = 4, 50, 5, 41
    tag = "B%d_L%d_M%d_seed%d" % (B, L, M, seed)
    b = weights.synthetic_xlnet_batch(B, L, 47, 74, seed=seed)
    ids, vis, aco, mask, seg = (torch.from_numpy(b[k]) for k in ("input_ids", "visual", "acoustic", "input_mask", "segment_ids"))
    tm = torch.from_numpy(g["target_mapping/tm/" + tag].astype(np.float32))
    pm = torch.from_numpy(g["target_mapping/perm/" + tag].astype(np.float32))
    assert tuple(tm.shape) == (B, M, L) and float(tm.sum()) == B * M
    m = X.load_deterministic(X.MAG_XLNetForSequenceClassification(X.XLNetConfigLite(), X.MultimodalConfig(1.0, 0.5), 47, 74)).eval()
    with torch.no_grad():
        out_g = m.transformer(ids, vis, aco, mask, seg, perm_mask=pm, target_mapping=tm)
        logits = m(ids, vis, aco, mask, seg, perm_mask=pm, target_mapping=tm)[0]
    assert len(m.transformer.hidden_g) == 13
    np.testing.assert_allclose(out_g.numpy(), g["target_mapping/output_g/" + tag], atol=2e-5)
    np.testing.assert_allclose(logits.numpy(), g["target_mapping/logits/" + tag], atol=2e-5)
