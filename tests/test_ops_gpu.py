"""GPU: operator-level parity of the HIP kernels (through the C ABI) against plain fp32/fp64 PyTorch on CPU.

Tolerances (written here, used below):
  fp32 parity mode : |err| <= 2e-5 * max|ref| + 1e-6      (exact-fp32 MFMA; only the summation order differs)
  bf16 perf mode   : inputs are rounded to bf16 first, reference computed in fp32 on the rounded inputs;
                     |err| <= 1.2e-2 * max|ref|            (one bf16 rounding of the output, bf16 intermediates)
"""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from bert_multimodal_transformer_amd import _lib, rng

DEV = "cuda:0"
DTS = [(_lib.DT_F32, torch.float32), (_lib.DT_BF16, torch.bfloat16)]


def tol(dt, ref):
    s = float(ref.abs().max())
    return (2e-5 * s + 1e-6) if dt == _lib.DT_F32 else (1.2e-2 * s + 1e-6)


def close(got, ref, dt, what="", scale=1.0):
    err = float((got.detach().cpu().double() - ref.double()).abs().max())
    t = tol(dt, ref) * scale
    assert err <= t, "%s: max|err| %.3e > tol %.3e (max|ref| %.3e)" % (what, err, t, float(ref.abs().max()))


def close_grad(got, ref, dt, what="", scale=1.0):
    """weight/bias gradients are sums over thousands of tokens: in bf16 mode a relu/clamp decision that flips on a
    near-zero pre-activation moves single elements by O(1) terms, so the bf16 bound is on the relative Frobenius
    error (<= 6e-2 ~ sqrt(fraction of flipped gates)); fp32 keeps the element-wise bound."""
    if dt == _lib.DT_F32:
        return close(got, ref, dt, what, scale)
    g, r = got.detach().cpu().double(), ref.double()
    rel = float((g - r).norm() / (r.norm() + 1e-30))
    assert rel <= 6e-2 * scale, "%s: relative Frobenius error %.3e" % (what, rel)


def rnd(shape, seed, tdt, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    x = (torch.rand(shape, generator=g) * 2 - 1) * scale
    return x.to(tdt).float()          # value exactly representable in the compute dtype


def stream():
    return torch.cuda.current_stream().cuda_stream


def gemm(dt, tdt, layout, epi, M, N, K, A, B, bias=None, R=None, alpha=1.0, drop=None, splits=1, tile=0, Cf_init=None):
    L = _lib.lib()
    Ad, Bd = A.to(DEV, tdt).contiguous(), B.to(DEV, tdt).contiguous()
    C1 = torch.zeros(M, N, dtype=tdt, device=DEV)
    C2 = torch.zeros(M, N, dtype=tdt, device=DEV)
    Cf = (Cf_init.to(DEV).clone() if Cf_init is not None else torch.zeros(M, N, dtype=torch.float32, device=DEV))
    bd = None if bias is None else bias.to(DEV)
    Rd = None if R is None else R.to(DEV, tdt).contiguous()
    _lib.check(L.mb_gemm(dt, layout, epi, M, N, K, _lib.ptr(Ad), Ad.shape[1], _lib.ptr(Bd), Bd.shape[1], _lib.ptr(C1), N,
                         _lib.ptr(C2), _lib.ptr(Cf), _lib.ptr(bd), _lib.ptr(Rd), N, alpha,
                         C.byref(drop) if drop is not None else None, splits, tile, stream()))
    torch.cuda.synchronize()
    return C1.float().cpu(), C2.float().cpu(), Cf.cpu()


@pytest.mark.parametrize("dt,tdt", DTS)
@pytest.mark.parametrize("M,N,K,tile", [(150, 192, 128, 64), (2400, 768, 768, 0), (300, 256, 3072, 128), (48, 768, 768, 64),
                                        (2400, 3072, 768, 0), (2400, 2304, 768, 256), (300, 136, 192, 256),   # 256 = eight-wave 256 x 128 tiles (bf16; fp32 falls back to 128)
                                        # 12872 = eight-wave 128 x 64 tiles, 128 k per stage (bf16; what it cannot take runs 64 x 64)
                                        (2400, 768, 3072, 12872), (2400, 768, 768, 12872), (300, 200, 384, 12872), (130, 64, 512, 12872),
                                        # 25672 = eight-wave 256 x 64 tiles, 64 k per stage (the same launches at T = 4096)
                                        (4096, 768, 3072, 25672), (4096, 768, 768, 0), (600, 200, 192, 25672), (260, 64, 320, 25672)])
def test_gemm_nt_epilogues(dt, tdt, M, N, K, tile):
    A, B = rnd((M, K), 1, tdt), rnd((N, K), 2, tdt, 0.1)     # asymmetric operands: a transposed C-write cannot pass
    bias = rnd((N,), 3, torch.float32)
    R = rnd((M, N), 4, tdt)
    ref = A.double() @ B.double().t()
    c, _, _ = gemm(dt, tdt, _lib.GEMM_NT, _lib.EPI_BIAS, M, N, K, A, B, bias=bias, alpha=0.5, tile=tile)
    close(c, (0.5 * ref + bias.double()).float(), dt, "bias")
    d, g, _ = gemm(dt, tdt, _lib.GEMM_NT, _lib.EPI_BIAS_GELU, M, N, K, A, B, bias=bias, tile=tile)
    uref = (ref + bias.double())
    dref = 0.5 * (1 + torch.erf(uref / 2 ** 0.5)) + uref * torch.exp(-0.5 * uref * uref) / (2 * np.pi) ** 0.5
    close(d, dref.float(), dt, "gelu.dgelu(u)")        # C = gelu'(acc + bias): what the backward's EPI_DGELU multiplies by
    close(g, torch.nn.functional.gelu(uref.float()), dt, "gelu.g", 2.0)
    key = _lib.make_dropkey(11, 3, 17, 0.1)
    mask = torch.from_numpy(rng.keep_mult(M * N, rng.make_key(11, 3, 17, 0.1))).view(M, N)
    c, _, _ = gemm(dt, tdt, _lib.GEMM_NT, _lib.EPI_BIAS_DROP_RES, M, N, K, A, B, bias=bias, R=R, drop=key, tile=tile)
    close(c, ((ref + bias.double()) * mask.double() + R.double()).float(), dt, "bias_drop_res")
    _, _, cf = gemm(dt, tdt, _lib.GEMM_NT, _lib.EPI_BIAS_F32, M, N, K, A, B, bias=bias, tile=tile)
    close(cf, (ref + bias.double()).float(), _lib.DT_F32 if dt == _lib.DT_F32 else dt, "bias_f32")


@pytest.mark.parametrize("dt,tdt", DTS)
@pytest.mark.parametrize("M,N,K,tile", [(150, 192, 128, 64), (2400, 768, 3072, 0), (2400, 3072, 768, 128), (2400, 768, 2304, 64),
                                        (100, 128, 64, 128), (2400, 768, 768, 128), (2400, 3072, 768, 0), (500, 384, 192, 256),
                                        (2400, 768, 3072, 12872), (2400, 768, 2304, 12872), (150, 192, 384, 12872), (2400, 768, 768, 12872),
                                        (4096, 768, 3072, 0), (4096, 768, 2304, 25672), (300, 192, 192, 25672)])
def test_gemm_nn_dgrad(dt, tdt, M, N, K, tile):
    A, B = rnd((M, K), 5, tdt), rnd((K, N), 6, tdt, 0.1)     # B stored [K][N]
    R = rnd((M, N), 7, tdt)
    ref = A.double() @ B.double()
    c, _, _ = gemm(dt, tdt, _lib.GEMM_NN, _lib.EPI_ADD_RES, M, N, K, A, B, R=R, tile=tile)
    close(c, (ref + R.double()).float(), dt, "add_res")
    c, _, cf = gemm(dt, tdt, _lib.GEMM_NN, _lib.EPI_DGELU, M, N, K, A, B, R=R, tile=tile)
    close(c, (ref * R.double()).float(), dt, "dgelu")      # R = gelu'(u) as saved by the forward's EPI_BIAS_GELU
    close(cf.view(-1)[:N], (ref * R.double()).sum(0).float(), dt, "dgelu.colsum")      # fused bias gradient: column sums of the fp32 values


@pytest.mark.parametrize("dt,tdt", DTS)
@pytest.mark.parametrize("M,N,K,splits,tile", [(192, 128, 150, 1, 64), (768, 768, 2400, 3, 64), (3072, 768, 2400, 2, 128),
                                               (1536, 64, 2400, 8, 64), (768, 768, 48, 1, 64),
                                               # K % 64 == 0 -> LDS-DMA ring + ds_read_b64_tr_b16 path
                                               (768, 768, 2432, 3, 64), (3072, 768, 2432, 2, 128), (768, 3072, 2432, 2, 128),
                                               (1536, 64, 2432, 8, 64), (2304, 768, 2432, 1, 0), (128, 128, 64, 1, 128),
                                               # 256 = the eight-wave ping-pong 256 x 128 tile (gemm_pp.hip; bf16): shortest k range (2 stages),
                                               # odd / even stage counts (the three-slot ring wraps differently), partial XCD regions
                                               (256, 128, 128, 1, 256), (768, 768, 2432, 1, 256), (768, 384, 2368, 1, 256), (3072, 768, 192, 1, 256)])
def test_gemm_tn_wgrad(dt, tdt, M, N, K, splits, tile):
    A, B = rnd((K, M), 8, tdt), rnd((K, N), 9, tdt, 0.1)     # both [K][rows]
    init = rnd((M, N), 10, torch.float32)
    ref = A.double().t() @ B.double() + init.double()
    _, _, cf = gemm(dt, tdt, _lib.GEMM_TN, _lib.EPI_ACCUM_F32, M, N, K, A, B, splits=splits, tile=tile, Cf_init=init)
    close(cf, ref.float(), _lib.DT_F32, "wgrad accumulate", 1.0 if dt == _lib.DT_F32 else 1.0)


@pytest.mark.parametrize("dt,tdt", DTS)
@pytest.mark.parametrize("tile", [64, 128, 256])
def test_gemm_grouped_wgrad(dt, tdt, tile):
    """the per-layer grouped weight-gradient launch == four independent fp32-accumulating TN GEMMs"""
    L = _lib.lib()
    K = 2432
    shapes = [(768, 3072), (3072, 768), (768, 768), (2304, 768)]          # [M_g = dY cols][N_g = X cols]
    dY = [rnd((K, m), 20 + i, tdt).to(DEV, tdt) for i, (m, n) in enumerate(shapes)]
    X = [rnd((K, n), 30 + i, tdt, 0.1).to(DEV, tdt) for i, (m, n) in enumerate(shapes)]
    init = [rnd((m, n), 40 + i, torch.float32) for i, (m, n) in enumerate(shapes)]
    dW = [t.to(DEV).clone() for t in init]
    if tile == 256 and dt != _lib.DT_BF16:
        pytest.skip("the 256 x 128 ping-pong tile is a bf16 kernel")
    ia = lambda v: (C.c_int * 4)(*v)
    pa = lambda ts: (C.c_void_p * 4)(*[t.data_ptr() for t in ts])
    _lib.check(L.mb_gemm_grouped_wgrad(dt, 4, ia([m for m, n in shapes]), ia([n for m, n in shapes]), K, pa(dY),
                                       ia([m for m, n in shapes]), pa(X), ia([n for m, n in shapes]), pa(dW),
                                       ia([n for m, n in shapes]), tile, stream()))
    torch.cuda.synchronize()
    for i in range(4):
        ref = dY[i].double().t() @ X[i].double() + init[i].to(DEV).double()
        close(dW[i].cpu(), ref.float().cpu(), _lib.DT_F32, "grouped wgrad %d" % i, 1.0)
    # shapes that are not whole tiles are refused (the engine then uses the single-problem launches)
    bad = L.mb_gemm_grouped_wgrad(dt, 1, ia([100, 0, 0, 0]), ia([128, 0, 0, 0]), K, pa(dY), ia([100, 0, 0, 0]), pa(X),
                                  ia([128, 0, 0, 0]), pa(dW), ia([128, 0, 0, 0]), tile, stream())
    assert bad != 0


@pytest.mark.parametrize("dt,tdt", DTS)
def test_layernorm_forward_backward(dt, tdt):
    L = _lib.lib()
    rows, H = 301, 768
    x = rnd((rows, H), 1, tdt, 2.0).requires_grad_(True)
    gamma = (1 + rnd((H,), 2, torch.float32, 0.1)).requires_grad_(True)
    beta = rnd((H,), 3, torch.float32, 0.1).requires_grad_(True)
    dy = rnd((rows, H), 4, tdt)
    for eps in (1e-12, 1e-5):
        y = torch.nn.functional.layer_norm(x.double(), (H,), gamma.double(), beta.double(), eps)
        y.backward(dy.double())
        xd, dyd = x.detach().to(DEV, tdt), dy.to(DEV, tdt)
        gd, bd = gamma.detach().to(DEV), beta.detach().to(DEV)      # keep the device copies alive across the launches
        yd = torch.empty_like(xd)
        mean = torch.empty(rows, device=DEV); rstd = torch.empty(rows, device=DEV)
        _lib.check(L.mb_layernorm_forward(dt, _lib.ptr(xd), _lib.ptr(gd), _lib.ptr(bd),
                                          eps, _lib.ptr(yd), _lib.ptr(mean), _lib.ptr(rstd), rows, H, None, stream()))
        close(yd.float(), y.detach().float(), dt, "ln fwd")
        dx = torch.empty_like(xd); dxd = torch.empty_like(xd)
        dg = torch.zeros(H, device=DEV); db = torch.zeros(H, device=DEV); dbias = torch.zeros(H, device=DEV)
        key = _lib.make_dropkey(3, 1, 18, 0.1)
        _lib.check(L.mb_layernorm_backward(dt, _lib.ptr(dyd), _lib.ptr(xd), _lib.ptr(gd), _lib.ptr(mean),
                                           _lib.ptr(rstd), _lib.ptr(dx), _lib.ptr(dxd), _lib.ptr(dg), _lib.ptr(db),
                                           _lib.ptr(dbias), rows, H, None, C.byref(key), stream()))
        torch.cuda.synchronize()
        close(dx.float(), x.grad.float(), dt, "ln dx", 2.0)
        close(dg, gamma.grad.float(), dt, "ln dgamma", 2.0)
        close(db, beta.grad.float(), dt, "ln dbeta", 2.0)
        mask = torch.from_numpy(rng.keep_mult(rows * H, rng.make_key(3, 1, 18, 0.1))).view(rows, H)
        close(dxd.float(), (x.grad.float() * mask), dt, "ln dx_drop", 2.0)
        close(dbias, (x.grad.float() * mask).sum(0), dt, "ln dbias", 3.0)
        x.grad = None; gamma.grad = None; beta.grad = None


@pytest.mark.parametrize("dt,tdt", DTS)
def test_embeddings_forward_backward(dt, tdt):
    L = _lib.lib()
    B, S, H, vocab = 5, 23, 768, 1000
    g = torch.Generator().manual_seed(0)
    ids = torch.randint(1, vocab, (B, S), generator=g)
    ids[:, 0] = 101; ids[0, 10:] = 0; ids[3, 5:] = 0          # pad id 0: no gradient (padding_idx)
    seg = torch.zeros(B, S, dtype=torch.long); seg[1, 7:] = 1
    word = rnd((vocab, H), 1, torch.float32, 0.05).requires_grad_(True)
    pos = rnd((64, H), 2, torch.float32, 0.05).requires_grad_(True)
    typ = rnd((2, H), 3, torch.float32, 0.05).requires_grad_(True)
    gamma = (1 + rnd((H,), 4, torch.float32, 0.1)).requires_grad_(True)
    beta = rnd((H,), 5, torch.float32, 0.1).requires_grad_(True)
    dy = rnd((B, S, H), 6, tdt)
    e = torch.nn.functional.embedding(ids, word.double(), padding_idx=0) + pos.double()[:S][None] + typ.double()[seg]
    y = torch.nn.functional.layer_norm(e, (H,), gamma.double(), beta.double(), 1e-12)
    y.backward(dy.double())
    _keep = {}

    def d(t):          # device copy kept alive until the end of the test (a temporary would be freed and reused)
        if id(t) not in _keep:
            _keep[id(t)] = t.detach().to(DEV)
        return _keep[id(t)]
    out = torch.empty(B * S, H, dtype=tdt, device=DEV)
    mean = torch.empty(B * S, device=DEV); rstd = torch.empty(B * S, device=DEV)
    idd, segd = ids.to(DEV), seg.to(DEV)
    _lib.check(L.mb_embed_forward(dt, _lib.ptr(idd), _lib.ptr(segd), _lib.ptr(d(word)), _lib.ptr(d(pos)), _lib.ptr(d(typ)),
                                  _lib.ptr(d(gamma)), _lib.ptr(d(beta)), 1e-12, _lib.ptr(out), _lib.ptr(mean), _lib.ptr(rstd),
                                  B, S, H, None, stream()))
    close(out.float().view(B, S, H), y.detach().float(), dt, "embed fwd")
    dws = torch.empty(B * S, H, device=DEV)
    gw = torch.zeros(vocab, H, device=DEV); gp = torch.zeros(64, H, device=DEV); gt = torch.zeros(2, H, device=DEV)
    gg = torch.zeros(H, device=DEV); gb = torch.zeros(H, device=DEV)
    dyd = dy.to(DEV, tdt).view(B * S, H).contiguous()
    _lib.check(L.mb_embed_backward(dt, _lib.ptr(dyd), _lib.ptr(idd), _lib.ptr(segd), _lib.ptr(d(word)), _lib.ptr(d(pos)),
                                   _lib.ptr(d(typ)), _lib.ptr(d(gamma)), _lib.ptr(mean), _lib.ptr(rstd), _lib.ptr(dws),
                                   _lib.ptr(gw), _lib.ptr(gp), _lib.ptr(gt), _lib.ptr(gg), _lib.ptr(gb), B, S, H, 0, None,
                                   stream()))
    torch.cuda.synchronize()
    s = 3.0
    close(gw, word.grad.float(), dt, "dword", s)
    close(gp, pos.grad.float(), dt, "dpos", s)
    close(gt, typ.grad.float(), dt, "dtype", s)
    close(gg, gamma.grad.float(), dt, "dgamma", s)
    close(gb, beta.grad.float(), dt, "dbeta", s)
    assert float(gw[0].abs().max()) == 0.0


def _attn_ref(qkv, mask, B, S, nh, pmask=None):
    H = nh * 64
    q, k, v = qkv.view(B, S, 3, nh, 64).permute(2, 0, 3, 1, 4)        # [B,nh,S,64]
    s = q @ k.transpose(-1, -2) / 8.0 + ((1.0 - mask[:, None, None, :].double()) * -10000.0)
    p = torch.softmax(s, -1)
    if pmask is not None:
        p = p * pmask
    return (p @ v).permute(0, 2, 1, 3).reshape(B * S, H)


@pytest.mark.parametrize("dt,tdt", DTS)
@pytest.mark.parametrize("B,S,p", [(3, 50, 0.0), (2, 128, 0.0), (2, 20, 0.0), (2, 50, 0.1), (2, 96, 0.1)])
def test_attention_forward_backward(dt, tdt, B, S, p):
    L = _lib.lib()
    nh, H = 12, 768
    qkv = rnd((B * S, 3 * H), 1, tdt, 2.0).requires_grad_(True)
    mask = torch.ones(B, S, dtype=torch.long)
    mask[0, S - 7:] = 0
    if B > 1:
        mask[1, 3:] = 0                                                  # nearly everything padded
    dctx = rnd((B * S, H), 2, tdt)
    pmask = None
    key = None
    if p > 0:
        key = _lib.make_dropkey(7, 5, 16, p)
        pmask = torch.from_numpy(rng.keep_mult(B * nh * S * S, rng.make_key(7, 5, 16, p))).view(B, nh, S, S).double()
    ctx = _attn_ref(qkv.double(), mask, B, S, nh, pmask)
    ctx.backward(dctx.double())
    qd = qkv.detach().to(DEV, tdt); md = mask.to(DEV)
    out = torch.zeros(B * S, H, dtype=tdt, device=DEV)
    kp = C.byref(key) if key is not None else None
    _lib.check(L.mb_attention_forward(dt, _lib.ptr(qd), _lib.ptr(md), _lib.ptr(out), B, S, nh, kp, stream()))
    close(out.float(), ctx.detach().float(), dt, "attn fwd", 2.0)
    dq = torch.zeros(B * S, 3 * H, dtype=tdt, device=DEV)
    dcd = dctx.to(DEV, tdt)
    _lib.check(L.mb_attention_backward(dt, _lib.ptr(qd), _lib.ptr(md), _lib.ptr(dcd), _lib.ptr(dq), B, S, nh, kp, stream()))
    torch.cuda.synchronize()
    close(dq.float(), qkv.grad.float(), dt, "attn bwd", 3.0)


def _mag_params(V, mode, dev):
    from oracle import weights
    shapes = {"W_hv.weight": (768, V + 768), "W_hv.bias": (768,), "W_ha.weight": (768, 74 + 768), "W_ha.bias": (768,),
              "W_v.weight": (768, V), "W_v.bias": (768,), "W_a.weight": (768, 74), "W_a.bias": (768,),
              "LayerNorm.weight": (768,), "LayerNorm.bias": (768,)}
    return {n: torch.from_numpy(weights.make_param("bert.MAG." + n, s, mode)).to(dev) for n, s in shapes.items()}


@pytest.mark.parametrize("V,mode,beta", [(47, "test", 1.0), (47, "init", 1.0), (47, "test", 1e-3), (35, "init", 1e-3),
                                         (35, "test", 1.0)])
def test_mag_module_matches_reference_golden(golden, V, mode, beta):
    """The reference's own MAG outputs (G1 fixture) vs the HIP MAG module, fp32 parity mode: <= 1e-3 (north_star);
    measured errors are ~1e-6.  Covers the hm_norm == 0 branch (mode=init) and the active clamp (beta=1)."""
    from oracle import weights
    from bert_multimodal_transformer_amd import MAG
    g = golden["g1_mag"]
    key = "V%d_%s_b%g" % (V, mode, beta)
    m = MAG(768, beta, 0.5, visual_dim=V, acoustic_dim=74).to(DEV).eval()
    m.load_state_dict(_mag_params(V, mode, DEV))
    b = weights.synthetic_bert_batch(2, 8, V, 74, seed=77, min_len=2)
    e = torch.tensor(weights.uniform("mag.e", (2, 8, 768), -1.5, 1.5), device=DEV, requires_grad=True)
    v = torch.tensor(b["visual"], device=DEV, requires_grad=True)
    a = torch.tensor(b["acoustic"], device=DEV, requires_grad=True)
    y = m(e, v, a)
    (y * torch.from_numpy(weights.uniform("mag.dy", tuple(y.shape))).to(DEV)).sum().backward()
    torch.cuda.synchronize()
    f32 = _lib.DT_F32
    close(y, torch.from_numpy(g[key + "/out"]), f32, "mag out", 5)
    close(e.grad, torch.from_numpy(g[key + "/d_text"]), f32, "mag d_text", 5)
    close(v.grad, torch.from_numpy(g[key + "/d_visual"]), f32, "mag d_visual", 5)
    close(a.grad, torch.from_numpy(g[key + "/d_acoustic"]), f32, "mag d_acoustic", 5)
    for n, p in m.named_parameters():
        assert torch.isfinite(p.grad).all()
        ref = torch.from_numpy(g[key + "/gslice/" + n])
        got = torch.from_numpy(weights.strided_sample(p.grad.cpu().numpy()))
        err = float((got - ref).abs().max())
        assert err <= 1e-4 * max(1.0, float(g[key + "/gnorm/" + n])), (n, err)
        np.testing.assert_allclose(float(p.grad.norm()), float(g[key + "/gnorm/" + n]), rtol=1e-3, atol=1e-5)


@pytest.mark.parametrize("cdt,dt", [(torch.float32, _lib.DT_F32), (torch.bfloat16, _lib.DT_BF16)])
def test_mag_module_vs_oracle_train_dropout_replay(cdt, dt):
    """T = 2400 tokens (config C2), dropout p = 0.5 ON, masks replayed in the oracle."""
    from oracle import mag_bert_ref as R, weights
    from bert_multimodal_transformer_amd import MAG, modeling
    torch.manual_seed(1234)
    V = 47
    m = MAG(768, 1.0, 0.5, visual_dim=V, acoustic_dim=74, compute_dtype=cdt).to(DEV).train()
    m.load_state_dict(_mag_params(V, "test", DEV))
    o = R.MAG(768, 1.0, 0.0, V, 74)
    o.load_state_dict({k: v.cpu() for k, v in m.state_dict().items()})
    b = weights.synthetic_bert_batch(48, 50, V, 74, seed=3)
    e0 = torch.from_numpy(weights.uniform("mag.e2", (48, 50, 768), -1.5, 1.5)).to(cdt).float()
    e = e0.clone().to(DEV).requires_grad_(True)
    y = m(e, torch.from_numpy(b["visual"]).to(DEV), torch.from_numpy(b["acoustic"]).to(DEV))
    step = modeling._STEP[0]
    mask = torch.from_numpy(rng.keep_mult(2400 * 768, rng.make_key(torch.initial_seed(), step, 1, 0.5))).view(48, 50, 768)
    eo = e0.clone().requires_grad_(True)
    yo = o(eo, torch.from_numpy(b["visual"]), torch.from_numpy(b["acoustic"])) * mask
    w = torch.from_numpy(weights.uniform("mag.dy2", (48, 50, 768))).to(cdt).float()
    (y * w.to(DEV)).sum().backward()
    (yo * w).sum().backward()
    torch.cuda.synchronize()
    close(y, yo.detach(), dt, "mag train out", 3)
    close(e.grad, eo.grad, dt, "mag train d_text", 4)
    for (n, p), (_, po) in zip(m.named_parameters(), o.named_parameters()):
        close_grad(p.grad, po.grad, dt, "mag grad " + n, 6 if dt == _lib.DT_F32 else 1)


@pytest.mark.parametrize("H", [256, 512, 1024])
def test_mag_module_other_hidden_sizes_vs_oracle(H):
    """MAG(hidden_size, ...) (modeling.py:7,22) away from TEXT_DIM = 768: output and every gradient against the oracle's MAG, fp32"""
    from oracle import mag_bert_ref as R, weights
    from bert_multimodal_transformer_amd import MAG
    V, A, B, L = 47, 74, 3, 20
    m = MAG(H, 1.0, 0.0, visual_dim=V, acoustic_dim=A).to(DEV).train()
    o = R.MAG(H, 1.0, 0.0, V, A).train()
    sd = {n: torch.from_numpy(weights.make_param("mag%d." % H + n, tuple(p.shape), "test")) for n, p in o.named_parameters()}
    o.load_state_dict(sd); m.load_state_dict({k: v.to(DEV) for k, v in sd.items()})
    b = weights.synthetic_bert_batch(B, L, V, A, seed=5)
    e0 = torch.from_numpy(weights.uniform("mag.eH%d" % H, (B, L, H), -1.5, 1.5))
    e = e0.clone().to(DEV).requires_grad_(True); eo = e0.clone().requires_grad_(True)
    w = torch.from_numpy(weights.uniform("mag.dyH%d" % H, (B, L, H)))
    y = m(e, torch.from_numpy(b["visual"]).to(DEV), torch.from_numpy(b["acoustic"]).to(DEV))
    yo = o(eo, torch.from_numpy(b["visual"]), torch.from_numpy(b["acoustic"]))
    (y * w.to(DEV)).sum().backward(); (yo * w).sum().backward()
    torch.cuda.synchronize()
    close(y, yo.detach(), _lib.DT_F32, "mag H=%d out" % H, 5)
    close(e.grad, eo.grad, _lib.DT_F32, "mag H=%d d_text" % H, 5)
    for (n, p), (_, po) in zip(m.named_parameters(), o.named_parameters()):
        close_grad(p.grad, po.grad, _lib.DT_F32, "mag H=%d grad %s" % (H, n), 6)


def test_adamw_kernel_matches_hf_formula():
    from oracle import optim_ref as O
    L = _lib.lib()
    n, nd = 4096 + 64, 4096
    g = torch.Generator().manual_seed(0)
    p0 = torch.randn(n, generator=g)
    po = torch.nn.Parameter(p0.clone()[:nd]); pn = torch.nn.Parameter(p0.clone()[nd:])
    opt = O.AdamW([{"params": [po], "weight_decay": 0.01}, {"params": [pn], "weight_decay": 0.0}], lr=1e-3)
    p = p0.clone().to(DEV); m = torch.zeros(n, device=DEV); v = torch.zeros(n, device=DEV)
    sh = torch.zeros(n, dtype=torch.bfloat16, device=DEV)
    for t in range(1, 6):
        gr = torch.randn(n, generator=g) * (10.0 ** (t - 3))
        po.grad = gr[:nd].clone(); pn.grad = gr[nd:].clone()
        opt.step()
        gd = (gr * 4.0).to(DEV)            # grad_scale 0.25 models the 1/world_size of a 4-rank sum
        _lib.check(L.mb_adamw_step(_lib.ptr(p), _lib.ptr(gd), _lib.ptr(m), _lib.ptr(v), _lib.ptr(sh), n, nd, 1024, 2048, 1e-3,
                                   0.9, 0.999, 1e-6, 0.01, t, 1, 0.25, 1, stream()))
        torch.cuda.synchronize()
        ref = torch.cat([po.detach(), pn.detach()])
        assert float((p.cpu() - ref).abs().max()) <= 2e-6 * float(ref.abs().max())
        assert float(gd.abs().max()) == 0.0                              # zero_grad fused
        assert torch.equal(sh[1024:2048].float().cpu(), p[1024:2048].to(torch.bfloat16).float().cpu())
        assert float(sh[:1024].abs().max()) == 0.0 and float(sh[2048:].abs().max()) == 0.0
