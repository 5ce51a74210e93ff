// Classification head of MAG_BertForSequenceClassification (/root/reference/bert.py:304-322) fused with the
// driver's MSE loss (/root/reference/multimodal_driver.py:372-373):
//   pooled = tanh(z)  [z = h[:,0] Wp^T + bp comes from the GEMM]  ->  dropout(0.1)  ->  logits = . Wc^T + bc
//   loss = mean((logits - labels)^2)                                  num_labels == 1 (the driver's regression)
//   loss = mean_b(logsumexp(logits_b) - logits_b[label_b])            num_labels  > 1 (bert.py:318-320 / xlnet.py:519-522:
//          CrossEntropyLoss; labels[b] then holds the class index of sample b as a float)
// One wave per sample; H = 768 (3 chunks of 4 columns per lane).  Tiny, launch-latency bound.
#include <algorithm>
#include "kernels.h"

namespace mb {

// Atomics on ONE address serialise at the L2 (measured: 48 samples x 2 atomics = most of a 15 us launch), so the four waves of
// a block meet in LDS first: one loss atomic per block, one classifier-gradient atomic per column per block.
template <int CH>
__global__ void __launch_bounds__(256) head_fwd_kernel(const float* __restrict__ z, const float* __restrict__ Wc,
                                                       const float* __restrict__ bc, const float* __restrict__ labels,
                                                       float* __restrict__ pooled, float* __restrict__ logits, float* loss,
                                                       float* loss_run, int B, int nl, DropKey drop) {
    drop.resolve();
    constexpr int H = CH * 256;
    __shared__ float lsum[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x * 4 + wave;
    const bool valid = b < B;
    float mine = 0.f;
    if (valid) {
        f32x4 pd[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int col = (c * 64 + lane) * 4;
            f32x4 t = *(const f32x4*)(z + (size_t)b * H + col);
#pragma unroll
            for (int r = 0; r < 4; ++r) t[r] = tanhf(t[r]);
            *(f32x4*)(pooled + (size_t)b * H + col) = t;
            const uint32_t idx = (uint32_t)b * H + col;
#pragma unroll
            for (int r = 0; r < 4; ++r) pd[c][r] = t[r] * drop_mult(drop, idx + r);
        }
        // cross entropy (nl > 1): online log-sum-exp over the classes, kept by lane 0
        float mx = -3.0e38f, se = 0.f, tgt = 0.f;
        const int y = (labels && nl > 1) ? (int)labels[b] : -1;
        for (int k = 0; k < nl; ++k) {
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const f32x4 w = *(const f32x4*)(Wc + (size_t)k * H + (c * 64 + lane) * 4);
                const f32x4 t = pd[c] * w;
                s += (t[0] + t[1]) + (t[2] + t[3]);
            }
            s = wave_sum(s) + bc[k];
            if (lane == 0) {
                logits[(size_t)b * nl + k] = s;
                if (labels && nl == 1) {
                    const float d = s - labels[b];
                    mine += d * d / (float)B;
                } else if (labels) {
                    const float nm = fmaxf(mx, s);
                    se = se * __expf(mx - nm) + __expf(s - nm);
                    mx = nm;
                    if (k == y) tgt = s;
                }
            }
        }
        if (lane == 0 && labels && nl > 1) mine = (mx + __logf(se) - tgt) / (float)B;
    }
    if (labels == nullptr || (loss == nullptr && loss_run == nullptr)) return;      // uniform
    if (lane == 0) lsum[wave] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float t = (lsum[0] + lsum[1]) + (lsum[2] + lsum[3]);
        if (loss) atomicAdd(loss, t);
        if (loss_run) atomicAdd(loss_run, t);
    }
}

template <class T, int CH>
__global__ void __launch_bounds__(256) head_bwd_kernel(const float* __restrict__ dlogits, const float* __restrict__ logits,
                                                       const float* __restrict__ labels, float loss_scale,
                                                       const float* __restrict__ pooled, const float* __restrict__ Wc,
                                                       T* __restrict__ dz, float* dWc, float* dbc, int B, int nl,
                                                       DropKey drop, GradAcc acc, float* dbp, u32x4* __restrict__ zero_p,
                                                       size_t zero_n16, int nb) {
    // blocks [nb, gridDim.x): clear zero_p[0, zero_n16) -- the token-gradient buffer whose [CLS] / last-token rows the next launch
    // fills (one launch less per backward than a separate zero_fill)
    if ((int)blockIdx.x >= nb) {
        const u32x4 z = {0u, 0u, 0u, 0u};
        for (size_t i = (size_t)(blockIdx.x - nb) * 256 + threadIdx.x; i < zero_n16; i += (size_t)(gridDim.x - nb) * 256) zero_p[i] = z;
        return;
    }
    drop.resolve();
    constexpr int H = CH * 256;
    __shared__ float wsum[4][H];
    __shared__ float bsum[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x * 4 + wave;
    const bool valid = b < B;
    const int bb = valid ? b : 0;
    f32x4 dpd[CH], pl[CH], dm[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int col = (c * 64 + lane) * 4;
        dpd[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        pl[c] = *(const f32x4*)(pooled + (size_t)bb * H + col);
        const uint32_t idx = (uint32_t)bb * H + col;
#pragma unroll
        for (int r = 0; r < 4; ++r) dm[c][r] = drop_mult(drop, idx + r);
    }
    float lse = 0.f;
    int y = -1;
    if (valid && !dlogits && nl > 1) {          // cross entropy: d logit_k = (softmax_k - [k == label]) / B
        float mx = -3.0e38f;
        for (int k = 0; k < nl; ++k) mx = fmaxf(mx, logits[(size_t)b * nl + k]);
        float se = 0.f;
        for (int k = 0; k < nl; ++k) se += __expf(logits[(size_t)b * nl + k] - mx);
        lse = mx + __logf(se);
        y = (int)labels[b];
    }
    for (int k = 0; k < nl; ++k) {
        float dl = 0.f;
        if (valid) {
            if (dlogits) dl = dlogits[(size_t)b * nl + k];
            else if (nl == 1) dl = 2.0f * (logits[b] - labels[b]) / (float)B * loss_scale;
            else dl = (__expf(logits[(size_t)b * nl + k] - lse) - (k == y ? 1.0f : 0.0f)) / (float)B * loss_scale;
        }
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int col = (c * 64 + lane) * 4;
            dpd[c] += dl * *(const f32x4*)(Wc + (size_t)k * H + col);
            if (dWc) *(f32x4*)(&wsum[wave][col]) = dl * pl[c] * dm[c];
        }
        if (lane == 0) bsum[wave] = dl;
        if (dWc || dbc) {                 // uniform
            __syncthreads();
            if (dWc)
                for (int col = threadIdx.x; col < H; col += 256)
                    grad_add(acc, dWc + (size_t)k * H + col, (wsum[0][col] + wsum[1][col]) + (wsum[2][col] + wsum[3][col]));
            if (dbc && threadIdx.x == 0) grad_add(acc, dbc + k, (bsum[0] + bsum[1]) + (bsum[2] + bsum[3]));
            __syncthreads();
        }
    }
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int col = (c * 64 + lane) * 4;
        const f32x4 d = dpd[c] * dm[c] * (1.0f - pl[c] * pl[c]);       // (dpd == 0 for a wave without a sample)
        if (valid) store4(dz + (size_t)b * H + col, d);
        if (dbp) *(f32x4*)(&wsum[wave][col]) = d;
    }
    if (dbp) {                            // uniform: the pooler / summary bias gradient = column sums of dz (was a colsum launch)
        __syncthreads();
        for (int col = threadIdx.x; col < H; col += 256)
            grad_add(acc, dbp + col, (wsum[0][col] + wsum[1][col]) + (wsum[2][col] + wsum[3][col]));
    }
}

int head_forward(const float* z, const float* Wc, const float* bc, const float* labels, float* pooled, float* logits,
                 float* loss, float* loss_run, int B, int H, int nl, DropKey drop, hipStream_t st) {
    if (H != 768) return MB_ERR_SHAPE;
    if (B <= 0) return MB_OK;
    hipLaunchKernelGGL((head_fwd_kernel<3>), dim3((B + 3) / 4), dim3(256), 0, st, z, Wc, bc, labels, pooled, logits, loss,
                       loss_run, B, nl, drop);
    return (int)hipGetLastError();
}

int head_backward(int dtype, const float* dlogits, const float* logits, const float* labels, float loss_scale,
                  const float* pooled, const float* Wc, void* dz, float* dWc, float* dbc, int B, int H, int nl,
                  DropKey drop, hipStream_t st, GradAcc acc, float* dbp, void* zero_p, size_t zero_bytes) {
    if (H != 768) return MB_ERR_SHAPE;
    if (B <= 0) return MB_OK;
    if (!dlogits && !(logits && labels)) return MB_ERR_ARG;
    if (zero_bytes % 16 || ((uintptr_t)zero_p & 15)) return MB_ERR_SHAPE;
    const int nb = (B + 3) / 4;
    const size_t n16 = zero_p ? zero_bytes / 16 : 0;
    const int zb = n16 ? (int)std::min<size_t>((n16 + 1023) / 1024, 512) : 0;
    if (dtype == DT_BF16)
        hipLaunchKernelGGL((head_bwd_kernel<bf16, 3>), dim3(nb + zb), dim3(256), 0, st, dlogits, logits, labels,
                           loss_scale, pooled, Wc, (bf16*)dz, dWc, dbc, B, nl, drop, acc, dbp, (u32x4*)zero_p, n16, nb);
    else if (dtype == DT_F32)
        hipLaunchKernelGGL((head_bwd_kernel<float, 3>), dim3(nb + zb), dim3(256), 0, st, dlogits, logits, labels,
                           loss_scale, pooled, Wc, (float*)dz, dWc, dbc, B, nl, drop, acc, dbp, (u32x4*)zero_p, n16, nb);
    else return MB_ERR_DTYPE;
    return (int)hipGetLastError();
}

}  // namespace mb
