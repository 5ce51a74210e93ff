// Device-side building blocks shared by every kernel of the MAG-BERT / MAG-XLNet hot path.
// gfx950 (CDNA4) only: 64-wide wavefronts, MFMA 16x16 tiles, 160 KB LDS per CU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

namespace mb {

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

enum { DT_F32 = 0, DT_BF16 = 1 };

// ---------------------------------------------------------------- scalar / vector conversions
__device__ __forceinline__ float to_f(float x) { return x; }
__device__ __forceinline__ float to_f(bf16 x) { return (float)x; }
template <class T> __device__ __forceinline__ T from_f(float x);
template <> __device__ __forceinline__ float from_f<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16 from_f<bf16>(float x) { return (bf16)x; }   // RNE (v_cvt_pk_bf16_f32)

__device__ __forceinline__ f32x4 load4(const float* p) { return *(const f32x4*)p; }
__device__ __forceinline__ f32x4 load4(const bf16* p) {
    bf16x4 v = *(const bf16x4*)p;
    f32x4 r = {(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
    return r;
}
__device__ __forceinline__ void store4(float* p, f32x4 v) { *(f32x4*)p = v; }
__device__ __forceinline__ void store4(bf16* p, f32x4 v) {
    bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
    *(bf16x4*)p = o;
}

// ---------------------------------------------------------------- counter-based dropout RNG
// keep(idx) is a pure function of (key, element index): forward and backward regenerate the same
// mask, nothing is stored.  tests/rng_ref.py holds the numpy twin used for mask replay in parity.
struct DropKey {
    uint32_t k0, k1;      // per-site, per-step key (host: splitmix64(seed, step, site))
    uint32_t thresh;      // drop if hash < thresh ; thresh = round(p * 2^32) ; 0 => dropout off
    float scale;          // 1/(1-p)
    // Replayed step graphs cannot carry a per-step value in their kernel arguments: there (k0, k1) of this site live in
    // device memory (written by step_prologue_kernel ahead of the graph launch) and every kernel fetches them once, at
    // entry, with two scalar loads.  nullptr = keys by value (eager launches, operator-level C ABI).
    const uint32_t* dyn;
    __device__ __forceinline__ void resolve() {
        if (dyn != nullptr && thresh != 0u) { k0 = dyn[0]; k1 = dyn[1]; }
    }
};
__device__ __forceinline__ uint32_t hash32(uint32_t idx, uint32_t k0, uint32_t k1) {
    uint32_t x = idx * 0x9E3779B1u + k0;
    x ^= x >> 16; x *= 0x85EBCA6Bu;
    x ^= x >> 13; x ^= k1; x *= 0xC2B2AE35u;
    x ^= x >> 16;
    return x;
}
// multiplier applied to element idx: 0 (dropped) or 1/(1-p) (kept)
__device__ __forceinline__ float drop_mult(const DropKey& d, uint32_t idx) {
    if (d.thresh == 0u) return 1.0f;
    return hash32(idx, d.k0, d.k1) < d.thresh ? 0.0f : d.scale;
}

// ---------------------------------------------------------------- gradient accumulation across workgroups
// Default: fp32 atomics -- the order in which the workgroups arrive is not reproducible, and one differently rounded sum can flip a
// bf16 rounding of the weight shadow.  Deterministic mode (MB_DETERMINISTIC=1): the same adds go, as 2^44-scaled 64-bit
// integers, into a shadow accumulator parallel to the gradient buffer; integer addition is associative, so any arrival order gives
// the same bits, and one conversion pass folds the shadow into the fp32 gradients before they are read (grad_fold).  Range +-5e5,
// resolution 5.7e-14 per addend.
struct GradAcc {
    long long* shadow;       // [n] parallel to base, or null = plain fp32 atomics
    const float* base;       // first element of the gradient buffer the shadow mirrors
};
constexpr float kGradFix = 17592186044416.0f;       // 2^44
__device__ __forceinline__ void grad_add(const GradAcc& a, float* dst, float v) {
    if (a.shadow) atomicAdd((unsigned long long*)(a.shadow + (dst - a.base)), (unsigned long long)__float2ll_rn(v * kGradFix));
    else atomicAdd(dst, v);
}

// ---------------------------------------------------------------- reductions
// Sum over the 64 lanes, returned in every lane.  DPP adds (plain VALU): an inclusive scan inside each 16-lane row (row_shr
// 1, 2, 4, 8), the row totals chained with row_bcast15 / row_bcast31, the grand total read from lane 63 -- seven instructions
// instead of six ds_bpermute round trips through the LDS crossbar (the row kernels do one or two of these per row on their
// critical path).
__device__ __forceinline__ float wave_sum(float v) {
#define MB_DPP(x, ctrl, rmask, bc) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), ctrl, rmask, 0xf, bc))
    v += MB_DPP(v, 0x111, 0xf, true);       // row_shr:1
    v += MB_DPP(v, 0x112, 0xf, true);       // row_shr:2
    v += MB_DPP(v, 0x114, 0xf, true);       // row_shr:4
    v += MB_DPP(v, 0x118, 0xf, true);       // row_shr:8   -> lane 15 of every row holds the row total
    v += MB_DPP(v, 0x142, 0xa, false);      // row_bcast15 -> rows 1 and 3 add the total of the row before
    v += MB_DPP(v, 0x143, 0xc, false);      // row_bcast31 -> rows 2 and 3 add lane 31 (rows 0 + 1): lane 63 holds everything
#undef MB_DPP
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// Sum over the 16 lanes of a DPP row (lanes that share lane >> 4), valid in lane 15 of the row only: an inclusive scan with four
// v_add_f32 row_shr steps (out-of-row sources read 0) -- plain VALU, no LDS crossbar traffic like the ds_bpermute behind
// __shfl_xor.  For reductions whose result one lane per row writes out.
__device__ __forceinline__ float row16_sum_to_lane15(float v) {
#define MB_ROW_SHR(x, n) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x110 + (n), 0xf, 0xf, true))
    v += MB_ROW_SHR(v, 1);
    v += MB_ROW_SHR(v, 2);
    v += MB_ROW_SHR(v, 4);
    v += MB_ROW_SHR(v, 8);
#undef MB_ROW_SHR
    return v;
}

// ---------------------------------------------------------------- MFMA 16x16 tile primitive
// A "fragment" is the 16 bytes one lane reads from row (lane & 15), 16-byte chunk (lane >> 4) of a
// 64-byte k-slab of an LDS/global image stored [row][k] with k contiguous:
//   bf16 : 8 elements k = (lane>>4)*8 + j   -> one v_mfma_f32_16x16x32_bf16
//   fp32 : 4 elements k = (lane>>4)*4 + j   -> four v_mfma_f32_16x16x4_f32 (exact fp32 fma chain)
// ---- piggy-backed prefetch ------------------------------------------------------------------------------------------------
// In the step the GEMMs find their weights in HBM, not in the 256 MB Infinity Cache (AdamW's 3 GB sweep and ~600 MB of activations
// pass between two uses), and run 15 % slower than over cache-resident operands (tools/gemm_bench --nset 48 vs 6:
// profiles/r03_gemm_cold_vs_warm.txt).  The latency-bound row kernels in front of them have the memory system idle: each of their
// blocks touches its slice of the NEXT launches' weights (K 16-byte loads per thread, issued behind the kernel's own first loads,
// retired at its end), which leaves the lines in the memory-side cache for every XCD.  `sink` is never written (null): it only
// keeps the loads alive.
// A second region (p2 / bytes2) takes the loads the first one leaves over, ONE load per `stride2` bytes: a touch is enough to bring the
// whole line into the memory-side cache, and the consumer of that region (the attention backward: q | k | v and the context rows a
// layer saved ~2 ms earlier) only needs them out of HBM, not in any particular L2.
struct Prefetch { const void* p; size_t bytes; uint32_t* sink; const void* p2 = nullptr; size_t bytes2 = 0; uint32_t stride2 = 64; };
// `fallback`: 1 KB of readable memory (what a launch without a prefetch region loads instead: K cache hits per thread).  The loads are
// unconditional -- offsets past the region(s) wrap into the first 1 KB (spread cache hits; clamping them all to one line made a
// hot spot) -- because a load inside divergent control flow makes the compiler wait for it at the join.
template <int K>
__device__ __forceinline__ void prefetch_issue(const Prefetch& pf, const void* fallback, u32x4 (&v)[K]) {
    const char* base = pf.p != nullptr ? (const char*)pf.p : (const char*)fallback;
    const bool real = pf.p != nullptr && pf.bytes >= 1024;      // (a smaller region could not take the wrapped offsets below)
    const size_t n1 = real ? pf.bytes / 16 : (size_t)0;         // loads the first region takes
    const bool two = pf.p2 != nullptr && pf.bytes2 >= 16;
    const size_t n2 = two ? (pf.bytes2 - 16) / pf.stride2 + 1 : (size_t)0;
    const size_t nthr = (size_t)gridDim.x * blockDim.x, tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    __builtin_amdgcn_sched_barrier(0);           // every load the kernel issued so far stays in front of these ...
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const size_t j = (size_t)k * nthr + tid;
        const char* src = base + ((j * 16) & (size_t)1008);     // past the region(s): spread over the first 1 KB (cache hits)
        if (j < n1) src = base + j * 16;
        else if (j - n1 < n2) src = (const char*)pf.p2 + (j - n1) * pf.stride2;
        v[k] = *(const u32x4*)src;
    }
    __builtin_amdgcn_sched_barrier(0);           // ... and nothing that follows is scheduled in between
}
template <int K>
__device__ __forceinline__ void prefetch_retire(const Prefetch& pf, u32x4 (&v)[K]) {
    uint32_t a = 0u;
    // (the empty asm pins the first use of the loaded registers HERE: without it the compiler folds them right behind the loads and
    //  the kernel's own work waits for the prefetch -- loads return in order)
#pragma unroll
    for (int k = 0; k < K; ++k) asm volatile("" : "+v"(v[k]) : : "memory");
#pragma unroll
    for (int k = 0; k < K; ++k) a ^= v[k][0] ^ v[k][1] ^ v[k][2] ^ v[k][3];
    if (pf.sink != nullptr) *pf.sink = a;
}

// mma16(acc, x, y):  acc[r] += sum_k X[(lane>>4)*4 + r][k] * Y[lane & 15][k]
// where x / y are the fragments this lane loaded from images X / Y.
template <class T> struct Frag;
template <> struct Frag<bf16> { typedef bf16x8 type; };
template <> struct Frag<float> { typedef f32x4 type; };

__device__ __forceinline__ void mma16(f32x4& acc, bf16x8 x, bf16x8 y) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, acc, 0, 0, 0);
}
__device__ __forceinline__ void mma16(f32x4& acc, f32x4 x, f32x4 y) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x[0], y[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x[1], y[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x[2], y[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x[3], y[3], acc, 0, 0, 0);
}

// erf-GELU (transformers 3.0.2 ACT2FN["gelu"]: x * 0.5 * (1 + erf(x / sqrt(2)))) and its derivative.
// Both need Phi(x) = 0.5 (1 + erf(x / sqrt 2)) and phi-like e = exp(-x^2 / 2); erf comes from Abramowitz-Stegun 7.1.26,
// erf(z) = 1 - (a1 t + ... + a5 t^5) exp(-z^2), t = 1 / (1 + p z), |error| <= 1.5e-7 -- and with z = x / sqrt 2 its exponential
// IS e, so one v_exp + one v_rcp + a Horner chain serve gelu and gelu' together (libm's erff is two branchy ranges per call;
// these sit in the epilogues of the two widest GEMMs).  1.5e-7 absolute on Phi is below fp32 rounding of the products it
// enters, far inside the 1e-3 logit contract of the fp32 parity mode.
// Evaluated two elements per instruction (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32: packed fp32 runs at twice the scalar VALU
// rate; only the two transcendentals -- v_exp_f32, v_rcp_f32, no IEEE division -- and the sign transfer stay per element): the
// GELU epilogue of the [T x 3072] GEMM was 9 us of VALU time per launch in scalar form.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void gelu_pair(f32x2 x, f32x2& g, f32x2& dg) {
    const f32x2 ax = {fabsf(x.x), fabsf(x.y)};
    const f32x2 d = ax * 0.23164190455f + 1.0f;                 // 1 + p |x| / sqrt 2
    const f32x2 q = x * x * -0.72134752044f;                    // -x^2 / 2 * log2(e)
    const f32x2 e = {__builtin_amdgcn_exp2f(q.x), __builtin_amdgcn_exp2f(q.y)};
    const f32x2 t = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
    // 0.5 * (a1 t + ... + a5 t^5): the halved A&S coefficients
    f32x2 poly = t * 0.5307027145f + -0.7265760135f;
    poly = poly * t + 0.7107068705f;
    poly = poly * t + -0.142248368f;
    poly = poly * t + 0.127414796f;
    const f32x2 h = poly * t * e;                               // 0.5 erfc(|x| / sqrt 2)
    const f32x2 u = 0.5f - h;                                   // Phi = 0.5 + sign(x) (0.5 - h)
    const f32x2 su = {__builtin_copysignf(u.x, x.x), __builtin_copysignf(u.y, x.y)};
    const f32x2 Phi = su + 0.5f;
    g = x * Phi;
    dg = x * 0.39894228040143268f * e + Phi;
}

}  // namespace mb
