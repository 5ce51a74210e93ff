"""numpy twin of the device dropout hash (csrc/common.h hash32 / engine.hip make_key) -- used by the parity tests
to replay the exact device masks inside the CPU oracle ("mask replay")."""
import numpy as np

SITE_EMB, SITE_MAG, SITE_HEAD, SITE_LAYER0 = 0, 1, 2, 16           # MAG-BERT: layer l sites 16 + 4l + {0 probs, 1 attn out, 2 ffn out}
# MAG-XLNet (csrc/xlnet_engine.hip): 0 word embedding [B,L,H], 1 MAG [B,L,H], 2 summary last_dropout [B,H], 3 final output
# [B,L,H], 4 pos_emb [2L,B,H]; layer l sites 16 + 8l + {0 probs [B,nh,L,L], 1 attn out [B,L,H], 2 ff act [B,L,d_inner], 3 ff out}
XS_EMB, XS_MAG, XS_HEAD, XS_FINAL, XS_POS, XS_LAYER0 = 0, 1, 2, 3, 4, 16
_M64 = (1 << 64) - 1


def _splitmix64(x):
    x = (x + 0x9E3779B97F4A7C15) & _M64
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & _M64
    return x ^ (x >> 31)


def make_key(seed, step, site, p):
    """(k0, k1, thresh, scale) exactly as engine.hip make_key."""
    if not p > 0:
        return 0, 0, 0, 1.0
    h = _splitmix64(_splitmix64(seed & _M64) ^ _splitmix64((step * 0x100000001B3 + site) & _M64))
    t = min(float(np.float32(p)) * 4294967296.0, 4294967295.0)
    return h & 0xFFFFFFFF, h >> 32, int(t + 0.5), float(np.float32(1.0) / (np.float32(1.0) - np.float32(p)))


def keep_mult(n, key):
    """fp32 multipliers (0 or 1/(1-p)) for element indices 0..n-1."""
    k0, k1, thresh, scale = key
    if thresh == 0:
        return np.ones(n, np.float32)
    x = (np.arange(n, dtype=np.uint64) * np.uint64(0x9E3779B1) + np.uint64(k0)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x85EBCA6B)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(13)
    x ^= np.uint64(k1)
    x = (x * np.uint64(0xC2B2AE35)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(16)
    return np.where(x < np.uint64(thresh), np.float32(0), np.float32(scale)).astype(np.float32)
