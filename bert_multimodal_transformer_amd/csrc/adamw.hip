// transformers==3.0.2 AdamW (/root/reference/multimodal_driver.py:28,345,384) as ONE launch over flat buffers.
//   m <- b1 m + (1-b1) g ; v <- b2 v + (1-b2) g^2 ; p <- p - step_size * m / (sqrt(v) + eps) ; p <- p - lr*wd*p
// (eps OUTSIDE the sqrt and before bias correction, decoupled decay AFTER the update on the updated p --
//  not torch.optim.AdamW).  HBM-bound: reads p,g,m,v and writes p,m,v = 28 B/param, + 4 B/param to zero the
//  gradient in place (replaces optimizer.zero_grad()) + 2 B/param for the bf16 operand shadow of GEMM weights.
// The flat layout puts every weight-decayed tensor first: elements [0, n_decay) decay, the rest do not.
#include <cstdlib>
#include "kernels.h"
#include "adamw_dev.h"

namespace mb {

// Variant kernel for experiments (MB_ADAMW_VAR): UNR quads in flight per thread, CHUNK: every block owns one contiguous range
template <bool NT, int UNR, bool CHUNK>
__global__ void __launch_bounds__(256) adamw_var_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, bf16* __restrict__ shadow, size_t n4, size_t n_decay,
                                                        size_t sh_begin, size_t sh_end, size_t keep_begin, size_t keep_end, AdamArgs a,
                                                        const AdamArgs* __restrict__ dyn, int zero_grad) {
    if (dyn) a = *dyn;
    const float omb1 = 1.0f - a.beta1, omb2 = 1.0f - a.beta2;
    const float decay = a.lr * a.weight_decay;
    size_t begin, end, stride;
    if constexpr (CHUNK) {
        const size_t per = ((n4 + gridDim.x - 1) / gridDim.x + 255) / 256 * 256;
        begin = (size_t)blockIdx.x * per + threadIdx.x; end = min(n4, (size_t)(blockIdx.x + 1) * per); stride = 256;
    } else {
        begin = (size_t)blockIdx.x * 256 + threadIdx.x; end = n4; stride = (size_t)gridDim.x * 256;
    }
    for (size_t i4 = begin; i4 < end; i4 += stride * UNR) {
        AdamQuad q[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u)
            if (i4 + u * stride < end) q[u] = adam_load<NT>(p, g, m, v, (i4 + u * stride) * 4);
#pragma unroll
        for (int u = 0; u < UNR; ++u)
            if (i4 + u * stride < end)
                adam_update_store<NT>(q[u], p, g, m, v, shadow, (i4 + u * stride) * 4, a, omb1, omb2, decay, n_decay, sh_begin, sh_end, keep_begin,
                                      keep_end, zero_grad);
    }
}

template <bool NT>
__global__ void __launch_bounds__(256) adamw_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, bf16* __restrict__ shadow, size_t n4,
                                                    size_t n_decay, size_t sh_begin, size_t sh_end, AdamArgs a,
                                                    const AdamArgs* __restrict__ dyn, int zero_grad) {
    if (dyn) a = *dyn;                  // replayed step graph: this step's lr / step size / gradient scale live in device memory
    const float omb1 = 1.0f - a.beta1, omb2 = 1.0f - a.beta2;
    const float decay = a.lr * a.weight_decay;
    for (size_t i4 = (size_t)blockIdx.x * 256 + threadIdx.x; i4 < n4; i4 += (size_t)gridDim.x * 256) {
#pragma clang fp contract(off)
        const size_t i = i4 * 4;
        f32x4 pv, gv, mv, vv;
        if constexpr (NT) {          // streamed once per step, never re-read before it is rewritten: keep it out of the caches
            pv = __builtin_nontemporal_load((const f32x4*)(p + i)); gv = __builtin_nontemporal_load((const f32x4*)(g + i));
            mv = __builtin_nontemporal_load((const f32x4*)(m + i)); vv = __builtin_nontemporal_load((const f32x4*)(v + i));
        } else {
            pv = *(const f32x4*)(p + i); gv = *(const f32x4*)(g + i); mv = *(const f32x4*)(m + i); vv = *(const f32x4*)(v + i);
        }
        gv *= a.grad_scale;
        mv = a.beta1 * mv + omb1 * gv;
        vv = a.beta2 * vv + omb2 * gv * gv;
#pragma unroll
        for (int r = 0; r < 4; ++r) pv[r] -= a.step_size * (mv[r] / (sqrtf(vv[r]) + a.eps));
        if (i < n_decay && decay > 0.f) pv -= decay * pv;
        if constexpr (NT) {
            __builtin_nontemporal_store(pv, (f32x4*)(p + i));
            __builtin_nontemporal_store(mv, (f32x4*)(m + i));
            __builtin_nontemporal_store(vv, (f32x4*)(v + i));
            if (zero_grad) __builtin_nontemporal_store(f32x4{0.f, 0.f, 0.f, 0.f}, (f32x4*)(g + i));
        } else {
            *(f32x4*)(p + i) = pv;
            *(f32x4*)(m + i) = mv;
            *(f32x4*)(v + i) = vv;
            if (zero_grad) *(f32x4*)(g + i) = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (shadow && i >= sh_begin && i < sh_end) store4(shadow + i, pv);
    }
}

// scalar tail (n % 4 elements) -- only hit by stand-alone tensors, the engine's flat buffers are 64-aligned
__global__ void adamw_tail_kernel(float* p, float* g, float* m, float* v, size_t begin, size_t n, size_t n_decay, AdamArgs a,
                                  const AdamArgs* dyn, int zero_grad) {
    if (dyn) a = *dyn;
    const size_t i = begin + threadIdx.x;
    if (i >= n) return;
    {
#pragma clang fp contract(off)
    const float gv = g[i] * a.grad_scale;
    const float mv = a.beta1 * m[i] + (1.0f - a.beta1) * gv;
    const float vv = a.beta2 * v[i] + (1.0f - a.beta2) * gv * gv;
    float pv = p[i] - a.step_size * (mv / (sqrtf(vv) + a.eps));
    if (i < n_decay && a.lr * a.weight_decay > 0.f) pv -= a.lr * a.weight_decay * pv;
    p[i] = pv; m[i] = mv; v[i] = vv;
    if (zero_grad) g[i] = 0.f;
    }
}

int adamw_step(float* p, float* g, float* m, float* v, void* shadow, size_t n, size_t n_decay, size_t sh_begin,
               size_t sh_end, AdamArgs a, int zero_grad, hipStream_t st, const AdamArgs* dyn, size_t keep_begin, size_t keep_end) {
    if (n == 0) return MB_OK;
    {   // MB_GEMM_LOG=1 (bench.py's in-run trace): how many parameters this sweep launch covers -- with riders (kernels.h AdamRide) the sweep at
        // the end of a step is no longer "all of them"
        static int log = -1;
        if (log < 0) { const char* e = getenv("MB_GEMM_LOG"); log = e ? atoi(e) : 0; }
        if (log) fprintf(stderr, "[magbert adamw] n=%zu\n", n);
    }
    if ((n_decay % 4 && n_decay < n) || (sh_begin % 4) || (sh_end % 4) || (keep_begin % 4) || (keep_end % 4)) return MB_ERR_SHAPE;
    if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) return MB_ERR_SHAPE;
    const size_t n4 = n / 4;
    if (n4) {
        unsigned grid = (unsigned)((n4 + 255) / 256);
        if (grid > 256 * 16) grid = 256 * 16;
        static int nt = -1;          // MB_ADAMW_NT=0: plain loads / stores (non-temporal measured 1.1 % faster per step: 4.86 vs 4.91 ms)
        if (nt < 0) { const char* e = getenv("MB_ADAMW_NT"); nt = e ? atoi(e) : 1; }
        // MB_ADAMW_VAR: 3 (default) = every block owns ONE contiguous range of each of the seven streams and keeps two quads in
        // flight (round 3, tools/adamw_bench: 5.38 vs 5.10 TB/s for the grid-strided kernel); 0 = that grid-strided kernel;
        // 1 / 2 / 4 / 5 = the other combinations measured (profiles/r03_adamw_variants.txt).  MB_ADAMW_GRID caps the grid.
        static int var = -1, vgrid = 0;
        if (var < 0) { const char* e = getenv("MB_ADAMW_VAR"); var = e ? atoi(e) : 3; const char* g2 = getenv("MB_ADAMW_GRID"); vgrid = g2 ? atoi(g2) : 0; }
        if (vgrid > 0 && (unsigned)vgrid < grid) grid = (unsigned)vgrid;
        const size_t kb = keep_begin, ke = keep_end;
        const int use = (var == 0 && ke > kb) ? 3 : var;       // (the grid-strided kernel has no keep range; this call only)
        // MB_ADAMW_NT selects non-temporal accesses in every variant
#define MB_AV(U, C) do { if (nt) hipLaunchKernelGGL((adamw_var_kernel<true, U, C>), dim3(grid), dim3(256), 0, st, p, g, m, v, (bf16*)shadow, n4, n_decay, sh_begin, sh_end, kb, ke, a, dyn, zero_grad); \
                         else hipLaunchKernelGGL((adamw_var_kernel<false, U, C>), dim3(grid), dim3(256), 0, st, p, g, m, v, (bf16*)shadow, n4, n_decay, sh_begin, sh_end, kb, ke, a, dyn, zero_grad); } while (0)
        if (use == 1) MB_AV(2, false); else if (use == 2) MB_AV(1, true); else if (use == 3) MB_AV(2, true); else if (use == 4) MB_AV(4, false); else if (use == 5) MB_AV(4, true); else
#undef MB_AV
        if (nt) hipLaunchKernelGGL(adamw_kernel<true>, dim3(grid), dim3(256), 0, st, p, g, m, v, (bf16*)shadow, n4, n_decay, sh_begin,
                                   sh_end, a, dyn, zero_grad);
        else hipLaunchKernelGGL(adamw_kernel<false>, dim3(grid), dim3(256), 0, st, p, g, m, v, (bf16*)shadow, n4, n_decay, sh_begin,
                                sh_end, a, dyn, zero_grad);
    }
    if (n % 4) hipLaunchKernelGGL(adamw_tail_kernel, dim3(1), dim3(64), 0, st, p, g, m, v, n4 * 4, n, n_decay, a, dyn, zero_grad);
    return (int)hipGetLastError();
}

}  // namespace mb
