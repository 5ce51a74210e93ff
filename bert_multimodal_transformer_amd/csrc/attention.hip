// LDS-resident fused self-attention for the BERT encoder, forward and backward (dh = 64, L <= 128).
//
// Replaces BertSelfAttention's score bmm + scale + mask add + softmax + dropout + context bmm + permutes
// (transformers 3.0.2, reached from /root/reference/bert.py:221-229):
//     S = Q K^T / sqrt(64) + (1 - mask) * -10000 ;  P = dropout(softmax(S)) ;  ctx = P V
// One workgroup per (batch, head).  Q/K/V (and dO in the backward) are read straight out of the token-major
// fused-QKV GEMM output [T][3H] (128-byte rows per head) into [row][d] LDS images; scores never leave the CU.
// Every product is built from the 16x16 MFMA tile primitive (common.h):
//   * a 16-byte "natural" fragment when the reduction index is contiguous in the image,
//   * a "k-major" fragment (strided element reads) when the operand is consumed transposed (V in P.V,
//     K in dS.K, dO / Q in the key-side products) -- attention is ~1% of the layer's FLOPs, so no transposed
//     copies are kept.
// The accumulators are laid out so a lane owns 4 consecutive columns of one row: row max / sum are an in-lane
// reduction plus two cross-lane steps (xor 16, 32).
// Backward is the two-sweep flash form with recomputation (no P is stored by the forward):
//   sweep A (query strips): recompute P, dP = dO V^T, D_i = sum_j dP_ij P_ij, dS -> dQ ; stash m_i, 1/l_i, D_i
//   sweep B (key strips)  : recompute P^T from the stashed row statistics -> dV = Pd^T dO, dK = dS^T Q
// Dropout masks are regenerated from the counter hash (common.h), index ((b*nh+h)*L + i)*L + j.
#include "attn_common.h"
#include "adamw_dev.h"

namespace mb {

// =============================================================================================== forward
template <class T, int LP, int NW>
__global__ void __launch_bounds__(NW * 64) attn_fwd_kernel(const T* __restrict__ qkv, const int64_t* __restrict__ mask,
                                                           T* __restrict__ ctx, float* __restrict__ probs,
                                                           const float* __restrict__ head_scale, int L, int nh,
                                                           DropKey drop) {
    drop.resolve();
    typedef AttnCfg<T> C;
    constexpr int PIT = C::ROWB + 16;                 // image pitch (bytes)
    constexpr int NT = LP / 16;                       // 16-wide tiles along L
    constexpr int DSL = 64 / C::SLAB;                 // k-slabs along d
    constexpr int LSL = LP / C::SLAB;                 // k-slabs along L
    typedef AccOp<T> AO;
    __shared__ __attribute__((aligned(16))) char smem[3 * LP * PIT + LP * 4];
    char* Qi = smem;
    char* Ki = smem + LP * PIT;
    char* Vi = smem + 2 * LP * PIT;
    float* mbias = (float*)(smem + 3 * LP * PIT);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x / nh, h = blockIdx.x % nh;
    const int H = nh * 64;
    const size_t ld = (size_t)3 * H;
    const T* base = qkv + (size_t)b * L * ld + h * 64;
    {
        char* const img[3] = {Qi, Ki, Vi};
        const T* const src[3] = {base, base + H, base + 2 * H};
        const size_t lds[3] = {ld, ld, ld};
        stage_heads<T, LP, NW * 64, 3>(img, PIT, src, lds, L);
    }
    for (int j = threadIdx.x; j < LP; j += NW * 64)
        mbias[j] = j < L ? (1.0f - (float)mask[(size_t)b * L + j]) * kMaskNeg : kPadNeg;
    __syncthreads();

    const float scale = 0.125f;
    // head_mask (bert.py:196-209 -> BertSelfAttention): the dropped probabilities of head h are multiplied by head_scale[h]
    const float hs = head_scale ? head_scale[h] : 1.0f;
    // every wave owns whole query strips from here on: no barrier, the probabilities never leave the registers (AccOp)
#pragma unroll 1
    for (int strip = wave; strip < NT; strip += NW) {
        f32x4 acc[NT];
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) {
            acc[jt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int sl = 0; sl < DSL; ++sl)
                mma16(acc[jt], frag_nat<T>(Ki, PIT, jt * 16 + (lane & 15), sl, lane),
                      frag_nat<T>(Qi, PIT, strip * 16 + (lane & 15), sl, lane));
        }
        // acc[jt][r] = S[i][j], i = strip*16 + (lane&15), j = jt*16 + (lane>>4)*4 + r
        float mx = -3.0e38f;
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) {
            const f32x4 mb4 = *(const f32x4*)(mbias + jt * 16 + (lane >> 4) * 4);
            acc[jt] = acc[jt] * scale + mb4;
            mx = fmaxf(mx, fmaxf(fmaxf(acc[jt][0], acc[jt][1]), fmaxf(acc[jt][2], acc[jt][3])));
        }
        mx = quad_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int jt = 0; jt < NT; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) { acc[jt][r] = __expf(acc[jt][r] - mx); sum += acc[jt][r]; }
        const float inv = 1.0f / quad_sum(sum);
        const int i = strip * 16 + (lane & 15);
        const uint32_t rowidx = ((uint32_t)blockIdx.x * L + (uint32_t)i) * L;
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) {
            const int j = jt * 16 + (lane >> 4) * 4;
            f32x4 p = acc[jt] * inv;
#pragma unroll
            for (int r = 0; r < 4; ++r) p[r] *= drop_mult(drop, rowidx + j + r) * hs;
            acc[jt] = p;                   // the dropped probabilities: operand of P.V below
            if (probs && i < L) {           // output_attentions (bert.py:147-151): the probabilities after dropout, fp32
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (j + r < L) probs[(size_t)rowidx + j + r] = p[r];
            }
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int sl = 0; sl < LSL; ++sl)
                mma16(o, AO::kmaj(Vi, PIT, sl, dt * 16 + (lane & 15), lane), AO::make(&acc[sl * AO::TILES]));
            // o[r] = ctx[i = strip*16 + (lane&15)][d = dt*16 + (lane>>4)*4 + r]
            if (i < L) store4(ctx + ((size_t)b * L + i) * H + h * 64 + dt * 16 + (lane >> 4) * 4, o);
        }
    }
}

// =============================================================================================== backward
#ifndef MB_ATTN_BWD_OCC
#define MB_ATTN_BWD_OCC 1          // waves per SIMD the eight-wave (L > 64) instantiation is compiled for (A/B builds: 4 = two workgroups per CU)
#endif
template <class T, int LP, int NW>
__device__ __forceinline__ void attn_bwd_body(const T* __restrict__ qkv, const int64_t* __restrict__ mask,
                                              const T* __restrict__ dctx, T* __restrict__ dqkv,
                                              float* __restrict__ dbias,
                                              const float* __restrict__ head_scale, int L, int nh, DropKey drop,
                                              unsigned long long* __restrict__ trace, GradAcc acc) {
    // MB_ATTN_TRACE=1: phase stamps of every block (100 MHz wall clock): 0 entry, 1 operands staged, 2 query sweep done,
    // 3 dQ bias flushed, 4 key sweep done, 5 exit
    auto stamp = [&](int k) { if (trace && threadIdx.x == 0) trace[(size_t)blockIdx.x * 8 + k] = wall_clock64(); };
    stamp(0);
    drop.resolve();
    typedef AttnCfg<T> C;
    typedef AccOp<T> AO;
    constexpr int PIT = C::ROWB + 16;
    constexpr int NT = LP / 16;
    constexpr int DSL = 64 / C::SLAB;
    constexpr int LSL = LP / C::SLAB;
    // four [row][d] images + the row vectors; the three column-sum tiles of the final flush reuse the Q image (dead by then)
    // ONE (L = 128, eight waves: every wave owns exactly one strip per sweep): the column sums of its dQ / dK / dV tiles (the fused QKV
    // bias gradient) go to LDS as soon as the tile exists instead of living in 48 accumulator registers until the end, and sweep B
    // walks the queries in two chunks (below): 222 -> 148 registers, 31.8 -> 30.4 us per launch at B = 32.  Capped at 128 registers
    // (-DMB_ATTN_BWD_OCC=4: two workgroups per CU, ONE round for the 384 workgroups of B = 32, 20 registers spilled) it is 29.7 us --
    // the launch was never two rounds of a fast kernel, two co-resident workgroups take twice as long each
    // (profiles/r06_attn_bwd128_ab.txt).  L <= 64 keeps the register accumulators (13.5 vs 13.8 us).
    constexpr bool ONE = (NT == NW) && LP > 64;
    constexpr int CSW = ONE ? 3 * NW * 64 * 4 : 0;
    __shared__ __attribute__((aligned(16))) char smem[4 * LP * PIT + 4 * LP * 4 + CSW];
    static_assert(LP * PIT >= 3 * NW * 64 * 4, "the Q image holds the three column-sum tiles of the flush");
    char* Qi = smem;
    char* Ki = smem + LP * PIT;
    char* Vi = smem + 2 * LP * PIT;
    char* Oi = smem + 3 * LP * PIT;          // dO image
    float* mbias = (float*)(smem + 4 * LP * PIT);
    float* rmax = mbias + LP;
    float* rinv = rmax + LP;
    float* rD = rinv + LP;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x / nh, h = blockIdx.x % nh;
    float* csw1 = (float*)(smem + 4 * LP * PIT + 4 * LP * 4);      // ONE: [3][NW][64] column sums, written per tile
    // column sums of one 16 x 64 output tile of this wave (rows >= L already zeroed by the caller) -> its slot of csw1
    auto colsum_tile = [&](int which, int dt, const f32x4& o) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float sred = row16_sum_to_lane15(o[r]);
            if ((lane & 15) == 15) csw1[(which * NW + wave) * 64 + dt * 16 + (lane >> 4) * 4 + r] = sred;
        }
    };
    const int H = nh * 64;
    const size_t ld = (size_t)3 * H;
    const T* base = qkv + (size_t)b * L * ld + h * 64;
#ifdef MB_ATTN_FUSE_PROBE
    // COST PROBE (measurement build only, scripts/gpu_r05d.sh; the numbers it produces are garbage): what would it cost this kernel to
    // compute its own dCtx tile -- dCtx[i][d] = sum_k dY[b*L + i][k] * Wo[k][h*64 + d], the out-projection's dgrad restricted to this
    // (sample, head) -- instead of reading it?  The favourable case: a transposed bf16 copy of Wo exists, so both
    // operands are "natural" fragments read straight from global memory (16 B per lane and k step); `dctx` plays dY (same [T][H]
    // shape), 64 rows of `qkv` play the [64][768] panel of Wo^T (same bytes, same reuse across the blocks of a head).
    if constexpr (sizeof(T) == 2 && LP == 64 && NW == 4) {
        {
            char* const img[3] = {Qi, Ki, Vi};
            const T* const src[3] = {base, base + H, base + 2 * H};
            const size_t lds[3] = {ld, ld, ld};
            stage_heads<T, LP, NW * 64, 3>(img, PIT, src, lds, L);
        }
        const int i = wave * 16 + (lane & 15);
        const T* arow = dctx + ((size_t)b * L + (i < L ? i : 0)) * H + (lane >> 4) * 8;          // this lane's dY row, its 8-element k chunk
        const T* wrow = qkv + (size_t)(h * 64 + (lane & 15)) * ld + (lane >> 4) * 8;             // "Wo^T" rows h*64 + dt*16 + (lane & 15)
        f32x4 o[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
        for (int ks = 0; ks < H / 32; ++ks) {
            const bf16x8 y = *(const bf16x8*)(arow + ks * 32);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) mma16(o[dt], *(const bf16x8*)(wrow + (size_t)dt * 16 * ld + ks * 32), y);
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {             // o[dt][r] = dCtx[i][dt*16 + (lane>>4)*4 + r] -> the dO image
            T* dst = (T*)(Oi + i * PIT) + dt * 16 + (lane >> 4) * 4;
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[r] = from_f<T>(i < L ? o[dt][r] : 0.f);
        }
    } else
#endif
    {
        char* const img[4] = {Qi, Ki, Vi, Oi};
        const T* const src[4] = {base, base + H, base + 2 * H, dctx + (size_t)b * L * H + h * 64};
        const size_t lds[4] = {ld, ld, ld, (size_t)H};
        stage_heads<T, LP, NW * 64, 4>(img, PIT, src, lds, L);
    }
    for (int j = threadIdx.x; j < LP; j += NW * 64)
        mbias[j] = j < L ? (1.0f - (float)mask[(size_t)b * L + j]) * kMaskNeg : kPadNeg;
    // per-lane running column sums of the dQ / dK / dV tiles this wave produces (its own row only; rows >= L excluded);
    // reduced over the 16 rows and the waves once, at the very end
    f32x4 cq[4], ck[4], cv[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) cq[dt] = ck[dt] = cv[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    // Fused QKV bias gradient: ONE reduction + one atomic per column per block, at the very end.  Flushed after each sweep, the
    // barrier behind every batch of atomics waited for their round trip to L2 -- 8 of the 25 us of a launch
    // (profiles/r02_attention_phases.txt).
    auto flush_all = [&]() {                 // called by every thread of the block (uniform)
        if (dbias == nullptr) return;
        __syncthreads();                     // every wave is done with the Q image (ONE: has written its column sums)
        float* csw = ONE ? csw1 : (float*)Qi;             // [3][NW][64]
        if constexpr (!ONE)
#pragma unroll
        for (int which = 0; which < 3; ++which) {
            f32x4 (&c4)[4] = which == 0 ? cq : which == 1 ? ck : cv;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float sred = row16_sum_to_lane15(c4[dt][r]);
                    if ((lane & 15) == 15) csw[(which * NW + wave) * 64 + dt * 16 + (lane >> 4) * 4 + r] = sred;
                }
        }
        stamp(6);
        if constexpr (!ONE) __syncthreads();
        for (int j = threadIdx.x; j < 192; j += NW * 64) {
            const int which = j >> 6, col = j & 63;
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) t += csw[(which * NW + w) * 64 + col];
            grad_add(acc, dbias + (size_t)which * H + h * 64 + col, t);
        }
    };
    __syncthreads();
    stamp(1);

    const float scale = 0.125f;
    // head_mask: ctx_h = head_scale[h] * (dropped P) V, so dQ, dK and dV of the head are the unmasked ones times head_scale[h]
    const float hs = head_scale ? head_scale[h] : 1.0f;
    T* dq_base = dqkv + (size_t)b * L * ld + h * 64;

    // ------------------------------------------------------------------ sweep A: query strips -> dQ, row stats
    // (a wave owns whole strips; dS stays in the accumulator registers and is the operand of dS.K -- AccOp -- so no barrier here)
#pragma unroll 1
    for (int strip = wave; strip < NT; strip += NW) {
        f32x4 sp[NT], dp[NT];
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) {
            sp[jt] = f32x4{0.f, 0.f, 0.f, 0.f};
            dp[jt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int sl = 0; sl < DSL; ++sl) {
                mma16(sp[jt], frag_nat<T>(Ki, PIT, jt * 16 + (lane & 15), sl, lane),
                      frag_nat<T>(Qi, PIT, strip * 16 + (lane & 15), sl, lane));
                mma16(dp[jt], frag_nat<T>(Vi, PIT, jt * 16 + (lane & 15), sl, lane),
                      frag_nat<T>(Oi, PIT, strip * 16 + (lane & 15), sl, lane));
            }
            // (L = 128: left alone the scheduler hoists the fragment loads of all eight tiles to the top -- 210 registers; fenced, the
            //  sweep lives in 128 = two workgroups per CU)
            if constexpr (LP > 64 && MB_ATTN_BWD_OCC > 1) __builtin_amdgcn_sched_barrier(0);
        }
        float mx = -3.0e38f;
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) {
            const f32x4 mb4 = *(const f32x4*)(mbias + jt * 16 + (lane >> 4) * 4);
            sp[jt] = sp[jt] * scale + mb4;
            mx = fmaxf(mx, fmaxf(fmaxf(sp[jt][0], sp[jt][1]), fmaxf(sp[jt][2], sp[jt][3])));
        }
        mx = quad_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int jt = 0; jt < NT; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) { sp[jt][r] = __expf(sp[jt][r] - mx); sum += sp[jt][r]; }
        const float inv = 1.0f / quad_sum(sum);
        const int i = strip * 16 + (lane & 15);
        const uint32_t rowidx = ((uint32_t)blockIdx.x * L + (uint32_t)i) * L;
        float dsum = 0.f;
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) {
            const int j = jt * 16 + (lane >> 4) * 4;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                sp[jt][r] *= inv;                                   // P_ij
                dp[jt][r] *= drop_mult(drop, rowidx + j + r);       // dP_ij (through the dropout)
                dsum += dp[jt][r] * sp[jt][r];
            }
        }
        const float D = quad_sum(dsum);
        if ((lane >> 4) == 0) { rmax[i] = mx; rinv[i] = inv; rD[i] = D; }
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) sp[jt] = sp[jt] * (dp[jt] - D) * scale;      // dS_ij
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int sl = 0; sl < LSL; ++sl)
                mma16(o, AO::kmaj(Ki, PIT, sl, dt * 16 + (lane & 15), lane), AO::make(&sp[sl * AO::TILES]));
            o = o * hs;
            if (i < L) store4(dq_base + (size_t)i * ld + dt * 16 + (lane >> 4) * 4, o);
            if constexpr (ONE) { if (dbias) colsum_tile(0, dt, i < L ? o : f32x4{0.f, 0.f, 0.f, 0.f}); }
            else if (i < L) cq[dt] += o;
        }
    }
    __syncthreads();          // the row statistics of every strip are in LDS
    stamp(2);
    stamp(3);

    // ------------------------------------------------------------------ sweep B: key strips -> dV, dK
    // The queries are swept in chunks of CHT 16-row tiles: dV and dK are sums over the queries, accumulated tile by tile in the same
    // order either way (same bits), but at L = 128 only 4 of the 8 tiles of S^T / dP^T are alive at a time -- 32 registers instead of 64.
    constexpr int CHT = (LP > 64 && ONE) ? 4 : NT;            // query tiles per chunk
    constexpr int NCH = NT / CHT, CSL = CHT / AO::TILES;      // chunks, k-slabs (along the queries) per chunk
    static_assert(NT % CHT == 0 && CHT % AO::TILES == 0, "whole chunks of whole slabs");
#pragma unroll 1
    for (int strip = wave; strip < NT; strip += NW) {
        const int j = strip * 16 + (lane & 15);          // this lane's key
        const float mbj = mbias[j];
        f32x4 ov[4], okk[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) ov[dt] = okk[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            f32x4 sp[CHT], dp[CHT];
#pragma unroll
            for (int il = 0; il < CHT; ++il) {
                const int it = ch * CHT + il;
                sp[il] = f32x4{0.f, 0.f, 0.f, 0.f};
                dp[il] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int sl = 0; sl < DSL; ++sl) {
                    mma16(sp[il], frag_nat<T>(Qi, PIT, it * 16 + (lane & 15), sl, lane),
                          frag_nat<T>(Ki, PIT, strip * 16 + (lane & 15), sl, lane));
                    mma16(dp[il], frag_nat<T>(Oi, PIT, it * 16 + (lane & 15), sl, lane),
                          frag_nat<T>(Vi, PIT, strip * 16 + (lane & 15), sl, lane));
                }
            }
            // sp[il][r] = S[i][j] (pre-scale), dp[il][r] = (dO V^T)[i][j],  i = it*16 + (lane>>4)*4 + r
#pragma unroll
            for (int il = 0; il < CHT; ++il) {
                const int i0 = (ch * CHT + il) * 16 + (lane >> 4) * 4;
                const f32x4 rm4 = *(const f32x4*)(rmax + i0), ri4 = *(const f32x4*)(rinv + i0), rd4 = *(const f32x4*)(rD + i0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = i0 + r;
                    const float p = __expf(sp[il][r] * scale + mbj - rm4[r]) * ri4[r];
                    const float dm = drop_mult(drop, ((uint32_t)blockIdx.x * L + (uint32_t)i) * L + (uint32_t)j);
                    sp[il][r] = p * (dp[il][r] * dm - rd4[r]) * scale;     // dS^T       -> dK
                    dp[il][r] = p * dm;                                    // dropped P^T -> dV
                }
            }
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int sl = 0; sl < CSL; ++sl)
                    mma16(ov[dt], AO::kmaj(Oi, PIT, ch * CSL + sl, dt * 16 + (lane & 15), lane), AO::make(&dp[sl * AO::TILES]));
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int sl = 0; sl < CSL; ++sl)
                    mma16(okk[dt], AO::kmaj(Qi, PIT, ch * CSL + sl, dt * 16 + (lane & 15), lane), AO::make(&sp[sl * AO::TILES]));
            if constexpr (NCH > 1) __builtin_amdgcn_sched_barrier(0);          // (one chunk's tiles alive at a time)
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const f32x4 o = ov[dt] * hs;
            if (j < L) store4(dq_base + (size_t)j * ld + 2 * H + dt * 16 + (lane >> 4) * 4, o);
            if constexpr (ONE) { if (dbias) colsum_tile(2, dt, j < L ? o : f32x4{0.f, 0.f, 0.f, 0.f}); }
            else if (j < L) cv[dt] += o;
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const f32x4 o = okk[dt] * hs;
            if (j < L) store4(dq_base + (size_t)j * ld + H + dt * 16 + (lane >> 4) * 4, o);
            if constexpr (ONE) { if (dbias) colsum_tile(1, dt, j < L ? o : f32x4{0.f, 0.f, 0.f, 0.f}); }
            else if (j < L) ck[dt] += o;
        }
    }
    stamp(4);
    flush_all();
    stamp(5);
}

template <class T, int LP, int NW>
__global__ void __launch_bounds__(NW * 64, (NW == 4 && LP <= 64) ? 2 : MB_ATTN_BWD_OCC) attn_bwd_kernel(const T* __restrict__ qkv, const int64_t* __restrict__ mask,
                                                           const T* __restrict__ dctx, T* __restrict__ dqkv,
                                                           float* __restrict__ dbias,
                                                           const float* __restrict__ head_scale, int L, int nh, DropKey drop,
                                                           unsigned long long* __restrict__ trace, GradAcc acc) {
    attn_bwd_body<T, LP, NW>(qkv, mask, dctx, dqkv, dbias, head_scale, L, nh, drop, trace, acc);
}
// The same launch with AdamW riders (kernels.h AdamRide) behind its `nblk` (batch, head) workgroups.  L <= 64: 576 workgroups in 1024
// slots, a latency-bound kernel with the memory pipes mostly idle; L = 128: one workgroup per CU, 384 of them = one and a half rounds --
// the riders get the 128 CUs the second round leaves empty.  A symbol of its own (the plain kernel is what profiles are keyed by).
template <class T, int LP, int NW>
__global__ void __launch_bounds__(NW * 64, (NW == 4 && LP <= 64) ? 2 : MB_ATTN_BWD_OCC) attn_bwd_ride_kernel(const T* __restrict__ qkv, const int64_t* __restrict__ mask,
                                                           const T* __restrict__ dctx, T* __restrict__ dqkv,
                                                           float* __restrict__ dbias,
                                                           const float* __restrict__ head_scale, int L, int nh, DropKey drop,
                                                           GradAcc acc, const AdamRide ride, int nblk) {
    if ((int)blockIdx.x >= nblk) {
        adam_ride_block<NW * 64, 2>(ride, (int)blockIdx.x - nblk);
        return;
    }
    attn_bwd_body<T, LP, NW>(qkv, mask, dctx, dqkv, dbias, head_scale, L, nh, drop, nullptr, acc);
}

// =============================================================================================== host
static unsigned long long* g_attn_trace = nullptr;      // MB_ATTN_TRACE=1 (measurement tooling): [blocks][8] stamps of the last backward
static int g_attn_trace_on = -1, g_attn_trace_blocks = 0;
static unsigned long long* attn_trace_buffer(int blocks) {
    if (g_attn_trace_on < 0) {
        const char* v = getenv("MB_ATTN_TRACE");
        g_attn_trace_on = v ? atoi(v) : 0;
        if (g_attn_trace_on && hipMalloc(&g_attn_trace, (size_t)8192 * 8 * sizeof(unsigned long long)) != hipSuccess) g_attn_trace_on = 0;
    }
    if (!g_attn_trace_on || blocks > 8192) return nullptr;
    g_attn_trace_blocks = blocks;
    return g_attn_trace;
}
int attention_trace_fetch(unsigned long long* host_out, int max_blocks) {
    if (!g_attn_trace || !host_out) return 0;
    const int n = g_attn_trace_blocks < max_blocks ? g_attn_trace_blocks : max_blocks;
    if (hipDeviceSynchronize() != hipSuccess) return 0;
    if (hipMemcpy(host_out, g_attn_trace, (size_t)n * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return 0;
    return n;
}
template <class T, int LP, int NW>
static int launch_fwd(const void* qkv, const int64_t* mask, void* ctx, float* probs, const float* hsc, int B, int L, int nh,
                      DropKey drop, hipStream_t st) {
    hipLaunchKernelGGL((attn_fwd_kernel<T, LP, NW>), dim3(B * nh), dim3(NW * 64), 0, st, (const T*)qkv, mask, (T*)ctx, probs, hsc,
                       L, nh, drop);
    return (int)hipGetLastError();
}
template <class T, int LP, int NW>
static int launch_bwd(const void* qkv, const int64_t* mask, const void* dctx, void* dqkv, float* dbias, const float* hsc, int B,
                      int L, int nh, DropKey drop, hipStream_t st, GradAcc acc, const AdamRide* ride) {
    if constexpr (sizeof(T) == 2) {
        if (ride != nullptr && ride->blocks > 0 && ride->n4 > 0) {
            gemm_log_ride(*ride);
            hipLaunchKernelGGL((attn_bwd_ride_kernel<T, LP, NW>), dim3(B * nh + ride->blocks), dim3(NW * 64), 0, st, (const T*)qkv, mask,
                               (const T*)dctx, (T*)dqkv, dbias, hsc, L, nh, drop, acc, *ride, B * nh);
            return (int)hipGetLastError();
        }
    }
    hipLaunchKernelGGL((attn_bwd_kernel<T, LP, NW>), dim3(B * nh), dim3(NW * 64), 0, st, (const T*)qkv, mask,
                       (const T*)dctx, (T*)dqkv, dbias, hsc, L, nh, drop, attn_trace_buffer(B * nh), acc);
    return (int)hipGetLastError();
}
// workgroups that fit the LAST round of a backward launch of nblk (batch, head) workgroups (bf16; 0 = no riders for this shape)
int attention_backward_free_slots(int dtype, int L, int nblk, int cus) {
    if (dtype != DT_BF16 || L < 1 || L > 128) return 0;
    const int LP = (L + 31) / 32 * 32;
    const int per_cu = LP <= 32 ? 3 : LP <= 64 ? 4 : 1;      // (registers / LDS of the bf16 ride instantiations: 148 | 126 registers, 19 | 38 KB; 57 | 82 KB: one block)
    const int slots = per_cu * cus;
    const int rounds = (nblk + slots - 1) / slots;
    return rounds * slots - nblk;
}

int attention_forward(int dtype, const void* qkv, const int64_t* mask, void* ctx, int B, int L, int nh, DropKey drop,
                      hipStream_t st, float* probs, const float* head_scale) {
    if (L < 1 || L > 128) return MB_ERR_SHAPE;
    const int LP = (L + 31) / 32 * 32;
    if (dtype == DT_BF16) {
        switch (LP) {
            case 32: return launch_fwd<bf16, 32, 2>(qkv, mask, ctx, probs, head_scale, B, L, nh, drop, st);
            case 64: return launch_fwd<bf16, 64, 4>(qkv, mask, ctx, probs, head_scale, B, L, nh, drop, st);
            case 96: return launch_fwd<bf16, 96, 4>(qkv, mask, ctx, probs, head_scale, B, L, nh, drop, st);
            default: return launch_fwd<bf16, 128, 4>(qkv, mask, ctx, probs, head_scale, B, L, nh, drop, st);
        }
    } else if (dtype == DT_F32) {
        switch (LP) {
            case 32: return launch_fwd<float, 32, 2>(qkv, mask, ctx, probs, head_scale, B, L, nh, drop, st);
            case 64: return launch_fwd<float, 64, 4>(qkv, mask, ctx, probs, head_scale, B, L, nh, drop, st);
            case 96: return launch_fwd<float, 96, 4>(qkv, mask, ctx, probs, head_scale, B, L, nh, drop, st);
            default: return launch_fwd<float, 128, 4>(qkv, mask, ctx, probs, head_scale, B, L, nh, drop, st);
        }
    }
    return MB_ERR_DTYPE;
}

int attention_backward(int dtype, const void* qkv, const int64_t* mask, const void* ctx, const void* dctx, void* dqkv,
                       float* dbias, int B, int L, int nh, DropKey drop, hipStream_t st, const float* head_scale, GradAcc acc, const AdamRide* ride) {
    (void)ctx;   // D_i is recomputed as sum_j dP_ij P_ij, the forward output is not needed
    if (L < 1 || L > 128) return MB_ERR_SHAPE;
    const int LP = (L + 31) / 32 * 32;
    if (dtype == DT_BF16) {
        switch (LP) {
            case 32: return launch_bwd<bf16, 32, 2>(qkv, mask, dctx, dqkv, dbias, head_scale, B, L, nh, drop, st, acc, ride);
            case 64: return launch_bwd<bf16, 64, 4>(qkv, mask, dctx, dqkv, dbias, head_scale, B, L, nh, drop, st, acc, ride);
            case 96: return launch_bwd<bf16, 96, 4>(qkv, mask, dctx, dqkv, dbias, head_scale, B, L, nh, drop, st, acc, ride);
            // 8 waves: all eight strips of a sweep at once, one block per CU (148 VGPRs since round 6, 82 KB of LDS; see ONE in the kernel).
            // Four waves with two strips each need 487 VGPRs, 217 of them spilled when capped at 256: not built.
            default: return launch_bwd<bf16, 128, 8>(qkv, mask, dctx, dqkv, dbias, head_scale, B, L, nh, drop, st, acc, ride);
        }
    } else if (dtype == DT_F32) {
        switch (LP) {
            case 32: return launch_bwd<float, 32, 2>(qkv, mask, dctx, dqkv, dbias, head_scale, B, L, nh, drop, st, acc, ride);
            case 64: return launch_bwd<float, 64, 4>(qkv, mask, dctx, dqkv, dbias, head_scale, B, L, nh, drop, st, acc, ride);
            case 96: return launch_bwd<float, 96, 2>(qkv, mask, dctx, dqkv, dbias, head_scale, B, L, nh, drop, st, acc, ride);
            default: return launch_bwd<float, 128, 2>(qkv, mask, dctx, dqkv, dbias, head_scale, B, L, nh, drop, st, acc, ride);   // 2 waves: LDS budget
        }
    }
    return MB_ERR_DTYPE;
}

}  // namespace mb
