"""Deterministic counter-hash tensors (TEST INFRASTRUCTURE).

Weights and synthetic batches are generated from (name, index) hashes so that the
GPU box, the build container and the golden generator all see bit-identical fp32
values without shipping 440 MB of weights (SURVEY.md section 8c, G-vectors).
numpy only; no torch RNG involved.
"""
import zlib
import numpy as np

_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_GOLD = np.uint64(0x9E3779B97F4A7C15)


def _mix64(x):
    """splitmix64 finaliser on a uint64 array (wraps mod 2**64)."""
    with np.errstate(over="ignore"):
        x = (x ^ (x >> np.uint64(30))) * _M1
        x = (x ^ (x >> np.uint64(27))) * _M2
        x = x ^ (x >> np.uint64(31))
    return x


def uniform(name, shape, lo=-1.0, hi=1.0, salt=0):
    """fp32 array of `shape`, element i = lo + (hi-lo) * u24(hash(name, salt, i))."""
    n = int(np.prod(shape)) if len(shape) else 1
    tid = np.uint64(zlib.crc32(name.encode()) & 0xFFFFFFFF)
    with np.errstate(over="ignore"):
        base = (tid << np.uint64(32)) ^ (np.uint64(salt) * _GOLD)
        ctr = np.arange(n, dtype=np.uint64) * _GOLD + base
    h = _mix64(ctr)
    u = (h >> np.uint64(40)).astype(np.float64) / float(1 << 24)   # 24-bit mantissa -> exact in fp32
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def randint(name, shape, lo, hi, salt=0):
    """int64 array in [lo, hi)."""
    n = int(np.prod(shape)) if len(shape) else 1
    tid = np.uint64(zlib.crc32(name.encode()) & 0xFFFFFFFF)
    with np.errstate(over="ignore"):
        base = (tid << np.uint64(32)) ^ (np.uint64(salt) * _GOLD)
        ctr = np.arange(n, dtype=np.uint64) * _GOLD + base
    h = _mix64(ctr)
    return (lo + (h % np.uint64(hi - lo)).astype(np.int64)).reshape(shape)


def make_param(name, shape, mode="test"):
    """Deterministic value for one parameter tensor, keyed by its state-dict name.

    mode="test":  weights U(-0.04,0.04) (std ~0.023, like the N(0,0.02) init law of
                  transformers PreTrainedModel._init_weights), biases U(-0.02,0.02),
                  LayerNorm gamma 1+U(-0.1,0.1), beta U(-0.05,0.05) -- every term is
                  non-trivial so a dropped bias / gamma shows up in parity.
    mode="init":  the reference's init law shape: biases 0, LayerNorm 1/0 (this is the
                  case that makes MAG's `hm_norm == 0` branch fire on zero-modality rows,
                  modeling.py:35-36).
    """
    leaf = name.split(".")[-1]
    is_ln = ("LayerNorm" in name) or ("layer_norm" in name)
    if is_ln and leaf == "weight":
        return np.ones(shape, np.float32) if mode == "init" else 1.0 + uniform(name, shape, -0.1, 0.1)
    if is_ln and leaf == "bias":
        return np.zeros(shape, np.float32) if mode == "init" else uniform(name, shape, -0.05, 0.05)
    if leaf == "bias" or leaf.endswith("_bias"):
        return np.zeros(shape, np.float32) if mode == "init" else uniform(name, shape, -0.02, 0.02)
    return uniform(name, shape, -0.04, 0.04)


def synthetic_bert_batch(B, L, V, A, seed=1234, vocab=30522, min_len=5):
    """Synthetic batch in the exact layout prepare_bert_input produces
    (multimodal_driver.py:143-173): ids=[101, tokens, 102, 0...], mask 1 on n+2 slots,
    segment ids all 0, modality rows EXACT zeros on [CLS]/[SEP]/pad rows."""
    tag = "batch%d" % seed
    n = randint(tag + ".len", (B,), min_len, L - 2 + 1)
    ids = np.zeros((B, L), np.int64)
    mask = np.zeros((B, L), np.int64)
    seg = np.zeros((B, L), np.int64)
    vis = uniform(tag + ".vis", (B, L, V), -2.0, 2.0)
    aco = uniform(tag + ".aco", (B, L, A), -2.0, 2.0)
    tok = randint(tag + ".tok", (B, L), 1000, vocab)
    for b in range(B):
        k = int(n[b])
        ids[b, 0] = 101
        ids[b, 1:1 + k] = tok[b, :k]
        ids[b, 1 + k] = 102
        mask[b, :k + 2] = 1
        vis[b, 0] = 0; vis[b, 1 + k:] = 0
        aco[b, 0] = 0; aco[b, 1 + k:] = 0
    label = uniform(tag + ".label", (B,), -3.0, 3.0)
    return dict(input_ids=ids, visual=vis, acoustic=aco, input_mask=mask, segment_ids=seg, label_ids=label)


def strided_sample(arr, n=32):
    """n evenly spaced elements of the flattened array (integer index arithmetic, exact)."""
    f = np.asarray(arr).reshape(-1)
    k = min(n, f.size)
    if k <= 1:
        return f[:k].astype(np.float32)
    idx = (np.arange(k, dtype=np.int64) * (f.size - 1)) // (k - 1)
    return f[idx].astype(np.float32)


def synthetic_xlnet_batch(B, L, V, A, seed=1234, vocab=32000, min_len=5):
    """Synthetic batch in the layout prepare_xlnet_input produces (multimodal_driver.py:176-205): LEFT padded,
    ids = [5(pad)..., tokens, 4(<sep>), 3(<cls>)], mask 0 on pads, segment ids 3 on pads / 0 on tokens+sep / 2 on cls,
    modality rows exact zeros on pad / sep / cls rows."""
    tag = "xbatch%d" % seed
    n = randint(tag + ".len", (B,), min_len, L - 2 + 1)
    ids = np.full((B, L), 5, np.int64)
    mask = np.zeros((B, L), np.int64)
    seg = np.full((B, L), 3, np.int64)
    vis = uniform(tag + ".vis", (B, L, V), -2.0, 2.0)
    aco = uniform(tag + ".aco", (B, L, A), -2.0, 2.0)
    tok = randint(tag + ".tok", (B, L), 10, vocab)
    for b in range(B):
        k = int(n[b])
        pad = L - k - 2
        ids[b, pad:pad + k] = tok[b, :k]
        ids[b, L - 2] = 4
        ids[b, L - 1] = 3
        mask[b, pad:] = 1
        seg[b, pad:L - 1] = 0
        seg[b, L - 1] = 2
        vis[b, :pad] = 0; vis[b, L - 2:] = 0
        aco[b, :pad] = 0; aco[b, L - 2:] = 0
    label = uniform(tag + ".label", (B,), -3.0, 3.0)
    return dict(input_ids=ids, visual=vis, acoustic=aco, input_mask=mask, segment_ids=seg, label_ids=label)
