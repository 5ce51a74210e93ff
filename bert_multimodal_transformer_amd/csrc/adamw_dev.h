// Device side of HF-AdamW (adamw.hip) shared with the kernels that carry an update inside their own launch (gemm.hip / gemm_pp.hip:
// AdamRide -- the optimizer update of the PREVIOUS layer's GEMM weights as extra workgroups of a layer's weight-gradient launch).
#pragma once
#include "kernels.h"

namespace mb {

// one quad of four consecutive parameters
struct AdamQuad { f32x4 p, g, m, v; };
template <bool NT> __device__ __forceinline__ AdamQuad adam_load(const float* p, const float* g, const float* m, const float* v, size_t i) {
    AdamQuad q;
    if constexpr (NT) {          // streamed once per step, never re-read before it is rewritten: keep it out of the caches
        q.p = __builtin_nontemporal_load((const f32x4*)(p + i)); q.g = __builtin_nontemporal_load((const f32x4*)(g + i));
        q.m = __builtin_nontemporal_load((const f32x4*)(m + i)); q.v = __builtin_nontemporal_load((const f32x4*)(v + i));
    } else {
        q.p = *(const f32x4*)(p + i); q.g = *(const f32x4*)(g + i); q.m = *(const f32x4*)(m + i); q.v = *(const f32x4*)(v + i);
    }
    return q;
}
template <bool NT> __device__ __forceinline__ void adam_update_store(AdamQuad q, float* p, float* g, float* m, float* v, bf16* shadow, size_t i,
                                                                     const AdamArgs& a, float omb1, float omb2, float decay, size_t n_decay,
                                                                     size_t sh_begin, size_t sh_end, size_t keep_begin, size_t keep_end, int zero_grad) {
    // No FMA contraction: the update is inlined into several kernels (the sweep, the riders of a weight-gradient launch), and left to the
    // compiler each context fuses different multiply-add pairs -- 1-ulp differences in m / v from the second step on (measured:
    // scripts/exp/ride_diag.py).  Unfused is also the reference's arithmetic: exp_avg.mul_(b1).add_(g, alpha=1-b1) rounds twice.
#pragma clang fp contract(off)
    q.g *= a.grad_scale;
    q.m = a.beta1 * q.m + omb1 * q.g;
    q.v = a.beta2 * q.v + omb2 * q.g * q.g;
#pragma unroll
    for (int r = 0; r < 4; ++r) q.p[r] -= a.step_size * (q.m[r] / (sqrtf(q.v[r]) + a.eps));
    if (i < n_decay && decay > 0.f) q.p -= decay * q.p;
    const bool zg = zero_grad && !(i >= keep_begin && i < keep_end);
    if constexpr (NT) {
        __builtin_nontemporal_store(q.p, (f32x4*)(p + i));
        __builtin_nontemporal_store(q.m, (f32x4*)(m + i));
        __builtin_nontemporal_store(q.v, (f32x4*)(v + i));
        if (zg) __builtin_nontemporal_store(f32x4{0.f, 0.f, 0.f, 0.f}, (f32x4*)(g + i));
    } else {
        *(f32x4*)(p + i) = q.p;
        *(f32x4*)(m + i) = q.m;
        *(f32x4*)(v + i) = q.v;
        if (zg) *(f32x4*)(g + i) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (shadow && i >= sh_begin && i < sh_end) store4(shadow + i, q.p);
}


// One rider workgroup (kernels.h AdamRide): workgroup `rb` of `r.blocks` owns one contiguous range of every stream.  A rider has a
// CU's memory pipeline to itself but only its own 4-8 waves to cover the HBM latency with, so the loop is a two-stage software
// pipeline: the loads of one half-iteration (UNR quads per stream and thread) are in flight while the other half is updated and
// stored.  Same arithmetic, same order as adamw_var_kernel (adam_update_store, contraction off): the same bits.
template <int NTHREADS, int UNR>
__device__ __forceinline__ void adam_ride_block(const AdamRide& r, int rb) {
    if (r.n4 == 0) return;
    const AdamArgs a = *r.dyn;
    const float omb1 = 1.0f - a.beta1, omb2 = 1.0f - a.beta2;
    const float decay = a.lr * a.weight_decay;
    const size_t per = ((r.n4 + r.blocks - 1) / r.blocks + NTHREADS - 1) / NTHREADS * NTHREADS;
    const size_t begin = (size_t)rb * per + threadIdx.x;
    const size_t end = r.n4 < (size_t)(rb + 1) * per ? r.n4 : (size_t)(rb + 1) * per;
    const size_t all = ~(size_t)0;
    constexpr size_t half = (size_t)NTHREADS * UNR;          // quads per half-iteration
    AdamQuad qa[UNR], qb[UNR];
    auto load = [&](AdamQuad (&q)[UNR], size_t i4) {
#pragma unroll
        for (int u = 0; u < UNR; ++u)
            if (i4 + (size_t)u * NTHREADS < end) q[u] = adam_load<true>(r.p, r.g, r.m, r.v, (i4 + (size_t)u * NTHREADS) * 4);
    };
    auto update = [&](AdamQuad (&q)[UNR], size_t i4) {
#pragma unroll
        for (int u = 0; u < UNR; ++u)
            if (i4 + (size_t)u * NTHREADS < end)
                adam_update_store<true>(q[u], r.p, r.g, r.m, r.v, r.shadow, (i4 + (size_t)u * NTHREADS) * 4, a, omb1, omb2, decay, all, 0, r.shadow ? all : 0,
                                        0, 0, r.zero_grad);
    };
    load(qa, begin);
    for (size_t i4 = begin; i4 < end; i4 += 2 * half) {
        load(qb, i4 + half);
        update(qa, i4);
        load(qa, i4 + 2 * half);
        update(qb, i4 + half);
    }
}

}  // namespace mb
