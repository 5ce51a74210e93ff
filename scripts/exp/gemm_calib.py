"""Calibration only (not product, not tests): our mb_gemm vs the vendor library (torch.matmul -> hipBLASLt) on the twelve
per-layer GEMM shapes of the bench workload, bf16, M = T = 2432.  Answers "how much headroom is left in the GEMM kernel"."""
import ctypes as C
import sys
import torch
sys.path.insert(0, ".")
from bert_multimodal_transformer_amd import _lib

DEV = "cuda:0"
L = _lib.lib()
T = 2432
bf = torch.bfloat16


def timeit(fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def ours(layout, epi, M, N, K, A, B, Cc, Cf=None, bias=None, R=None, tile=0):
    st = torch.cuda.current_stream().cuda_stream
    return lambda: _lib.check(L.mb_gemm(_lib.DT_BF16, layout, epi, M, N, K, _lib.ptr(A), A.shape[1], _lib.ptr(B), B.shape[1],
                                        _lib.ptr(Cc), N, _lib.ptr(Cc), _lib.ptr(Cf), _lib.ptr(bias), _lib.ptr(R), N, 1.0, None, 1,
                                        tile, st))


rows = []
for name, N, K in (("qkv", 2304, 768), ("out", 768, 768), ("ffn1", 3072, 768), ("ffn2", 768, 3072)):
    X = torch.randn(T, K, device=DEV, dtype=bf)
    W = torch.randn(N, K, device=DEV, dtype=bf) * 0.02
    Y = torch.empty(T, N, device=DEV, dtype=bf)
    dY = torch.randn(T, N, device=DEV, dtype=bf)
    dX = torch.empty(T, K, device=DEV, dtype=bf)
    dW = torch.zeros(N, K, device=DEV, dtype=torch.float32)
    dWb = torch.empty(N, K, device=DEV, dtype=bf)
    bias = torch.zeros(N, device=DEV)
    fl = 2.0 * T * N * K
    Wt = W.t()
    dYt = dY.t()
    for kind, fo, ft in (
        ("fwd   %-4s NT" % name, ours(_lib.GEMM_NT, _lib.EPI_BIAS, T, N, K, X, W, Y, bias=bias), lambda: torch.matmul(X, Wt, out=Y)),
        ("dgrad %-4s NN" % name, ours(_lib.GEMM_NN, _lib.EPI_ADD_RES, T, K, N, dY, W, dX, R=X),
         lambda: torch.matmul(dY, W, out=dX)),
        ("wgrad %-4s TN" % name, ours(_lib.GEMM_TN, _lib.EPI_ACCUM_F32, N, K, T, dY, X, dWb, Cf=dW), lambda: torch.matmul(dYt, X, out=dWb)),
    ):
        a, b = timeit(fo), timeit(ft)
        rows.append((kind, a, b))
        print("%s  ours %7.2f us (%6.1f TF/s)   hipBLASLt %7.2f us (%6.1f TF/s)" % (kind, a, fl / a / 1e6, b, fl / b / 1e6), flush=True)
# grouped wgrad: the four weight gradients of a layer in one launch
shapes = [(768, 3072), (3072, 768), (768, 768), (2304, 768)]
dYs = [torch.randn(T, m, device=DEV, dtype=bf) for m, n in shapes]
Xs = [torch.randn(T, n, device=DEV, dtype=bf) for m, n in shapes]
dWs = [torch.zeros(m, n, device=DEV) for m, n in shapes]
ia = lambda v: (C.c_int * 4)(*v)
pa = lambda ts: (C.c_void_p * 4)(*[t.data_ptr() for t in ts])
Ms, Ns = ia([m for m, n in shapes]), ia([n for m, n in shapes])
pY, pX, pW = pa(dYs), pa(Xs), pa(dWs)
fl4 = sum(2.0 * T * m * n for m, n in shapes)
for tile in (64, 128):
    st = torch.cuda.current_stream().cuda_stream
    t = timeit(lambda: _lib.check(L.mb_gemm_grouped_wgrad(_lib.DT_BF16, 4, Ms, Ns, T, pY, Ms, pX, Ns, pW, Ns, tile, st)))
    print("grouped wgrad (4 problems, tile %3d): %7.2f us (%6.1f TF/s)   [four launches: %.1f us]" %
          (tile, t, fl4 / t / 1e6, sum(r[1] for r in rows if r[0].startswith("wgrad"))), flush=True)
print("per layer: ours %.1f us, library %.1f us" % (sum(r[1] for r in rows), sum(r[2] for r in rows)))
