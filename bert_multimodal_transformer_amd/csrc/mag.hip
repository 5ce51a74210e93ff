// Multimodal Adaptation Gate -- /root/reference/modeling.py:6-51 -- as MI355X kernels.
//
// The four Linears of MAG (W_hv, W_ha on cat(modality, text); W_v, W_a on the modality) are regrouped by
// INPUT so that no concatenation is ever materialised (modeling.py:27-28 build two [T, ~830] cats):
//     Ze[T,2H] = e  . [W_hv[:, V:] ; W_ha[:, A:]]^T      (text part of both gates,      K = H)
//     Zv[T,2H] = vp . [W_hv[:, :V] ; W_v        ]^T      (visual part of gate_v | W_v v, K = Vp)
//     Za[T,2H] = ap . [W_ha[:, :A] ; W_a        ]^T      (acoustic part of gate_a | W_a a, K = Ap)
// three MFMA GEMMs (gemm.hip) on zero-padded modality operands vp/ap (coalesced loads of the (B,L,47|74)
// tensors happen once, in pack_pad).  This file holds the weight (un)packing and the fused row kernels:
//   forward : relu gates, h_m, the two row norms, hm==0 replacement, alpha=min(en/(hn+eps)*beta,1),
//             alpha*h+e, LayerNorm(eps 1e-5), dropout                    (modeling.py:27-49)
//   backward: the exact adjoint, recomputing the gate from the saved pre-activation panels, including the
//             where / min / norm sub-gradients at the hm==0 rows (zero, never NaN).
#include "kernels.h"
#include "mag_pack.h"

namespace mb {

template <class T>
__global__ void mag_pack_w_kernel(const float* __restrict__ W_hv, const float* __restrict__ W_ha,
                                  const float* __restrict__ W_v, const float* __restrict__ W_a, T* __restrict__ We,
                                  T* __restrict__ Wv, T* __restrict__ Wa, MagDims d) {
    mag_pack_w_range<T>(W_hv, W_ha, W_v, W_a, We, Wv, Wa, d, (size_t)blockIdx.x * 256 + threadIdx.x, (size_t)gridDim.x * 256);
}

__global__ void mag_unpack_g_kernel(const float* __restrict__ dWe, const float* __restrict__ dWv,
                                    const float* __restrict__ dWa, float* dW_hv, float* dW_ha, float* dW_v, float* dW_a,
                                    MagDims d) {
    const int H = d.H, V = d.V, A = d.A, Vp = d.Vp, Ap = d.Ap;
    const size_t n_hv = (size_t)H * (V + H), n_ha = (size_t)H * (A + H), n_v = (size_t)H * V, n_a = (size_t)H * A;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_hv + n_ha + n_v + n_a; i += (size_t)gridDim.x * 256) {
        if (i < n_hv) {
            const int j = (int)(i / (V + H)), c = (int)(i % (V + H));
            dW_hv[i] += c < V ? dWv[(size_t)j * Vp + c] : dWe[(size_t)j * H + (c - V)];
        } else if (i < n_hv + n_ha) {
            const size_t k = i - n_hv;
            const int j = (int)(k / (A + H)), c = (int)(k % (A + H));
            dW_ha[k] += c < A ? dWa[(size_t)j * Ap + c] : dWe[(size_t)(H + j) * H + (c - A)];
        } else if (i < n_hv + n_ha + n_v) {
            const size_t k = i - n_hv - n_ha;
            const int j = (int)(k / V), c = (int)(k % V);
            dW_v[k] += dWv[(size_t)(H + j) * Vp + c];
        } else {
            const size_t k = i - n_hv - n_ha - n_v;
            const int j = (int)(k / A), c = (int)(k % A);
            dW_a[k] += dWa[(size_t)(H + j) * Ap + c];
        }
    }
}

// per-lane gate state for one row (CH chunks of 4 columns)
template <class T, int CH>
struct GateRow {
    f32x4 e[CH], zgv[CH], zga[CH], pv[CH], pa[CH], h[CH];
    float en, hn, hn0, thr, alpha;

    __device__ __forceinline__ void compute(const T* __restrict__ ep, const T* __restrict__ Ze, const T* __restrict__ Zv,
                                            const T* __restrict__ Za, const float* b_hv, const float* b_ha,
                                            const float* b_v, const float* b_a, size_t row, int lane, float beta_shift) {
        constexpr int H = CH * 256;
        float en2 = 0.f, hn2 = 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int col = (c * 64 + lane) * 4;
            e[c] = load4(ep + row * H + col);
            zgv[c] = load4(Ze + row * 2 * H + col) + load4(Zv + row * 2 * H + col) + *(const f32x4*)(b_hv + col);
            zga[c] = load4(Ze + row * 2 * H + H + col) + load4(Za + row * 2 * H + col) + *(const f32x4*)(b_ha + col);
            pv[c] = load4(Zv + row * 2 * H + H + col) + *(const f32x4*)(b_v + col);
            pa[c] = load4(Za + row * 2 * H + H + col) + *(const f32x4*)(b_a + col);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float gv = fmaxf(zgv[c][r], 0.f), ga = fmaxf(zga[c][r], 0.f);     // modeling.py:27-28
                h[c][r] = gv * pv[c][r] + ga * pa[c][r];                               // modeling.py:30
                en2 += e[c][r] * e[c][r];
                hn2 += h[c][r] * h[c][r];
            }
        }
        en = sqrtf(wave_sum(en2));                       // modeling.py:32
        hn = sqrtf(wave_sum(hn2));                       // modeling.py:33
        hn0 = (hn == 0.f) ? 1.f : hn;                    // modeling.py:35-36
        thr = en / (hn0 + 1e-6f) * beta_shift;           // modeling.py:38
        alpha = fminf(thr, 1.f);                         // modeling.py:40-43
    }
};

template <class T, int CH>
__global__ void __launch_bounds__(256) mag_gate_fwd_kernel(const T* __restrict__ e, const T* __restrict__ Ze,
                                                           const T* __restrict__ Zv, const T* __restrict__ Za,
                                                           const float* b_hv, const float* b_ha, const float* b_v,
                                                           const float* b_a, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float ln_eps, float beta_shift,
                                                           T* __restrict__ out, float* mean, float* rstd, int rows,
                                                           DropKey drop) {
    drop.resolve();
    constexpr int H = CH * 256;
    constexpr float invH = 1.0f / H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    if (row >= rows) return;
    GateRow<T, CH> g;
    g.compute(e, Ze, Zv, Za, b_hv, b_ha, b_v, b_a, (size_t)row, lane, beta_shift);
    f32x4 s[CH];
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        s[c] = g.alpha * g.h[c] + g.e[c];                 // modeling.py:45,48
        sum += (s[c][0] + s[c][1]) + (s[c][2] + s[c][3]);
    }
    const float mu = wave_sum(sum) * invH;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float dd = s[c][r] - mu; q += dd * dd; }
    const float rs = 1.0f / sqrtf(wave_sum(q) * invH + ln_eps);
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int col = (c * 64 + lane) * 4;
        f32x4 o = (s[c] - mu) * rs * *(const f32x4*)(gamma + col) + *(const f32x4*)(beta + col);
        const uint32_t idx = (uint32_t)row * H + col;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] *= drop_mult(drop, idx + r);       // modeling.py:47-49
        store4(out + (size_t)row * H + col, o);
    }
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
}

template <class T, int CH, int RPW>
__global__ void __launch_bounds__(256, 2) mag_gate_bwd_kernel(const T* __restrict__ dout, const T* __restrict__ e,
                                                           const T* __restrict__ Ze, const T* __restrict__ Zv,
                                                           const T* __restrict__ Za, const float* b_hv, const float* b_ha,
                                                           const float* b_v, const float* b_a, const float* __restrict__ gamma,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           float beta_shift, T* __restrict__ de, T* __restrict__ dZe,
                                                           T* __restrict__ dZv, T* __restrict__ dZa, float* db_hv,
                                                           float* db_ha, float* db_v, float* db_a, float* dgamma,
                                                           float* dbeta, int rows, DropKey drop, GradAcc acc, float* part_a,
                                                           float* part_b) {
    drop.resolve();
    constexpr int H = CH * 256;
    constexpr float invH = 1.0f / H;
    __shared__ float lds[4 * 6 * H];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 part[6][CH];
#pragma unroll
    for (int q = 0; q < 6; ++q)
#pragma unroll
        for (int c = 0; c < CH; ++c) part[q][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int i = 0; i < RPW; ++i) {
        const int row = (blockIdx.x * 4 + wave) * RPW + i;
        if (row >= rows) break;
        GateRow<T, CH> g;
        g.compute(e, Ze, Zv, Za, b_hv, b_ha, b_v, b_a, (size_t)row, lane, beta_shift);
        const float mu = mean[row], rs = rstd[row];
        // ---- LayerNorm + dropout backward -> ds (grad wrt s = alpha*h + e)
        f32x4 ds[CH];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int col = (c * 64 + lane) * 4;
            f32x4 dy = load4(dout + (size_t)row * H + col);
            const uint32_t idx = (uint32_t)row * H + col;
#pragma unroll
            for (int r = 0; r < 4; ++r) dy[r] *= drop_mult(drop, idx + r);
            const f32x4 xh = ((g.alpha * g.h[c] + g.e[c]) - mu) * rs;
            const f32x4 dxh = dy * *(const f32x4*)(gamma + col);
            s1 += (dxh[0] + dxh[1]) + (dxh[2] + dxh[3]);
            const f32x4 t = dxh * xh;
            s2 += (t[0] + t[1]) + (t[2] + t[3]);
            part[4][c] += dy * xh;
            part[5][c] += dy;
            ds[c] = dxh;         // finished below once m1, m2 are known
        }
        const float m1 = wave_sum(s1) * invH, m2 = wave_sum(s2) * invH;
        float dal = 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const f32x4 xh = ((g.alpha * g.h[c] + g.e[c]) - mu) * rs;
            ds[c] = (ds[c] - m1 - xh * m2) * rs;
            const f32x4 t = ds[c] * g.h[c];
            dal += (t[0] + t[1]) + (t[2] + t[3]);
        }
        const float dalpha = wave_sum(dal);
        // ---- alpha = min(thr, 1) ; thr = en / (hn0 + eps) * beta ; hn0 = where(hn == 0, 1, hn)
        const float dthr = (g.thr < 1.f) ? dalpha : (g.thr == 1.f ? 0.5f * dalpha : 0.f);
        const float inv = 1.0f / (g.hn0 + 1e-6f);
        const float den = dthr * beta_shift * inv;
        const float dhn = (g.hn == 0.f) ? 0.f : -dthr * g.en * beta_shift * inv * inv;
        const float ce = (g.en > 0.f) ? den / g.en : 0.f;       // d||e||/de = e/||e|| (0 at the origin)
        const float ch = (g.hn > 0.f) ? dhn / g.hn : 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int col = (c * 64 + lane) * 4;
            f32x4 dE, dzgv, dzga, dpv, dpa;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float dh = g.alpha * ds[c][r] + ch * g.h[c][r];
                dE[r] = ds[c][r] + ce * g.e[c][r];
                const float gv = fmaxf(g.zgv[c][r], 0.f), ga = fmaxf(g.zga[c][r], 0.f);
                dzgv[r] = g.zgv[c][r] > 0.f ? dh * g.pv[c][r] : 0.f;
                dzga[r] = g.zga[c][r] > 0.f ? dh * g.pa[c][r] : 0.f;
                dpv[r] = dh * gv;
                dpa[r] = dh * ga;
            }
            store4(de + (size_t)row * H + col, dE);
            store4(dZe + (size_t)row * 2 * H + col, dzgv);
            store4(dZe + (size_t)row * 2 * H + H + col, dzga);
            store4(dZv + (size_t)row * 2 * H + col, dzgv);
            store4(dZv + (size_t)row * 2 * H + H + col, dpv);
            store4(dZa + (size_t)row * 2 * H + col, dzga);
            store4(dZa + (size_t)row * 2 * H + H + col, dpa);
            part[0][c] += dzgv;
            part[1][c] += dzga;
            part[2][c] += dpv;
            part[3][c] += dpa;
        }
    }
    // column sums -> bias / LayerNorm grads
    float* const dst[6] = {db_hv, db_ha, db_v, db_a, dgamma, dbeta};
#pragma unroll
    for (int q = 0; q < 6; ++q)
#pragma unroll
        for (int c = 0; c < CH; ++c) *(f32x4*)(lds + (wave * 6 + q) * H + (c * 64 + lane) * 4) = part[q][c];
    __syncthreads();
    for (int i = threadIdx.x; i < 6 * H; i += 256) {
        const int q = i / H, col = i % H;
        const float s = lds[(0 * 6 + q) * H + col] + lds[(1 * 6 + q) * H + col] + lds[(2 * 6 + q) * H + col] +
                        lds[(3 * 6 + q) * H + col];
        if (part_a != nullptr) {
            // this block's slab of the two partial sets ([block][3][H], the layout of ln_bwd's): no atomics here -- 300 blocks x
            // 4,608 columns onto the same 4,608 addresses were most of this launch; a ln_reduce launch sums the slabs
            (q < 3 ? part_a : part_b)[((size_t)blockIdx.x * 3 + q % 3) * H + col] = s;
        } else if (dst[q] != nullptr) {
            grad_add(acc, dst[q] + col, s);
        }
    }
}

#define MB_DISPATCH_T(dtype, ...)                                  \
    if ((dtype) == DT_BF16) { typedef bf16 T; __VA_ARGS__ }        \
    else if ((dtype) == DT_F32) { typedef float T; __VA_ARGS__ }   \
    else return MB_ERR_DTYPE;

int mag_pack_weights(int dtype, const float* W_hv, const float* W_ha, const float* W_v, const float* W_a, void* We,
                     void* Wv, void* Wa, MagDims d, hipStream_t st) {
    MB_DISPATCH_T(dtype, {
        hipLaunchKernelGGL((mag_pack_w_kernel<T>), dim3(1024), dim3(256), 0, st, W_hv, W_ha, W_v, W_a, (T*)We, (T*)Wv,
                           (T*)Wa, d);
    })
    return (int)hipGetLastError();
}

int mag_unpack_wgrads(const float* dWe, const float* dWv, const float* dWa, float* dW_hv, float* dW_ha, float* dW_v,
                      float* dW_a, MagDims d, hipStream_t st) {
    hipLaunchKernelGGL(mag_unpack_g_kernel, dim3(1024), dim3(256), 0, st, dWe, dWv, dWa, dW_hv, dW_ha, dW_v, dW_a, d);
    return (int)hipGetLastError();
}

int mag_gate_forward(int dtype, const void* e, const void* Ze, const void* Zv, const void* Za, const float* b_hv,
                     const float* b_ha, const float* b_v, const float* b_a, const float* gamma, const float* beta,
                     float ln_eps, float beta_shift, void* out, float* mean, float* rstd, MagDims d, DropKey drop,
                     hipStream_t st) {
    if (d.H % 256 || d.H < 256 || d.H > 1024) return MB_ERR_SHAPE;      // MAG(hidden_size, ...) (modeling.py:7,22): rows of 256 .. 1024
    if (d.T <= 0) return MB_OK;
#define MB_MAG_FWD(CHV) hipLaunchKernelGGL((mag_gate_fwd_kernel<T, CHV>), dim3((d.T + 3) / 4), dim3(256), 0, st, (const T*)e, (const T*)Ze, \
                           (const T*)Zv, (const T*)Za, b_hv, b_ha, b_v, b_a, gamma, beta, ln_eps, beta_shift, (T*)out, mean, rstd, d.T, drop)
    MB_DISPATCH_T(dtype, {
        switch (d.H / 256) { case 1: MB_MAG_FWD(1); break; case 2: MB_MAG_FWD(2); break; case 3: MB_MAG_FWD(3); break; default: MB_MAG_FWD(4); }
    })
#undef MB_MAG_FWD
    return (int)hipGetLastError();
}

int mag_gate_backward(int dtype, const void* dout, const void* e, const void* Ze, const void* Zv, const void* Za,
                      const float* b_hv, const float* b_ha, const float* b_v, const float* b_a, const float* gamma,
                      const float* mean, const float* rstd, float beta_shift, void* de, void* dZe, void* dZv, void* dZa,
                      float* db_hv, float* db_ha, float* db_v, float* db_a, float* dgamma, float* dbeta, MagDims d,
                      DropKey drop, hipStream_t st, GradAcc acc, float* part_a, float* part_b, int* nblk) {
    if (d.H % 256 || d.H < 256 || d.H > 1024) return MB_ERR_SHAPE;
    if ((part_a == nullptr) != (part_b == nullptr)) return MB_ERR_ARG;
    constexpr int RPW = 2;
    if (nblk) *nblk = (d.T + 4 * RPW - 1) / (4 * RPW);
    if (d.T <= 0) return MB_OK;
#define MB_MAG_BWD(CHV) hipLaunchKernelGGL((mag_gate_bwd_kernel<T, CHV, RPW>), dim3((d.T + 4 * RPW - 1) / (4 * RPW)), dim3(256), 0, st, \
                           (const T*)dout, (const T*)e, (const T*)Ze, (const T*)Zv, (const T*)Za, b_hv, b_ha, b_v, b_a, \
                           gamma, mean, rstd, beta_shift, (T*)de, (T*)dZe, (T*)dZv, (T*)dZa, db_hv, db_ha, db_v, db_a, \
                           dgamma, dbeta, d.T, drop, acc, part_a, part_b)
    MB_DISPATCH_T(dtype, {
        switch (d.H / 256) { case 1: MB_MAG_BWD(1); break; case 2: MB_MAG_BWD(2); break; case 3: MB_MAG_BWD(3); break; default: MB_MAG_BWD(4); }
    })
#undef MB_MAG_BWD
    return (int)hipGetLastError();
}

}  // namespace mb
