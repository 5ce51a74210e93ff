// XLNet relative attention core (transformers 3.0.2 XLNetRelativeAttention.rel_attn_core, reached from
// /root/reference/xlnet.py:374-385), forward and backward, L <= 128, head dim 64, one workgroup per (batch, head[, strip group]):
//     ac[i,j] = (q_i + r_w_bias) . k_j
//     bd[i,j] = (q_i + r_r_bias) . kr_{L-i+j}          (rel_shift folded into the index: no [L,2L] reshape tricks)
//     ef[i,j] = (q_i + r_s_bias) . seg_embed[seg_i != seg_j]
//     P = dropout(softmax((ac+bd+ef)/8 - 1e30 * [key j is padding and i != j])) ;  vec = P V
// q|k|v come token-major from three projections into one [T][3H] buffer, kr = (dropped sinusoid) . W_r is [B][2L][H].
// The biases are folded as per-column constants ((q+b).k = q.k + b.k), so one Q image serves all three terms.
// Unlike the BERT kernel the probabilities ARE written out (P, and G = dL/d(ac+bd+ef) in the backward, [B,nh,L,L],
// 2.9 MB/layer in bf16): the backward needs them in three index spaces (query-major, key-major, shifted position-major)
// and keeping three transposed copies plus five operand images in LDS does not fit 160 KB in fp32.
//   xl_attn_fwd    : scores -> softmax -> P (saved, undropped) -> vec
//   xl_attn_bwd_q  : dP, G (saved), dq = G.K + Gshift.KR + sum_j G.seg ; r_w/r_r/r_s bias and seg_embed gradients
//   xl_attn_bwd_kv : dv = Pd^T dO ; dk = G^T (q + r_w_bias) ; dkr[p] = sum_i G[i, p-L+i] (q_i + r_r_bias)
#include <cstdlib>
#include "attn_common.h"
#include "adamw_dev.h"

namespace mb {

constexpr float kXlMask = 1.0e30f;      // modeling_xlnet: attn_score - 1e30 * attn_mask (fp32)

struct XlParams {
    const float* r_w_bias; const float* r_r_bias; const float* r_s_bias;   // [nh][64]
    const float* seg_embed;                                                // [2][nh][64]
    const int64_t* seg; const int64_t* mask;                               // [B][L]
    const float* head_scale;                                               // [nh] or null: head_mask of this layer (xlnet.py:383)
    // [B][L][L] bytes or null: perm[b][i][j] != 0 <=> data_mask[i, j, b] > 0 (xlnet.py:265-286: input_mask[j] + perm_mask[i, j]),
    // query i may not attend to key j (the i == j exemption of non_tgt_mask, xlnet.py:288-296, still applies).  Forward only: the
    // backward works from the saved probabilities, which are exact zeros wherever a score was masked.
    const uint8_t* perm;
    int gstream;                                                           // forward only: 1 = the query stream's mask, the i == j exemption dropped (attn_mask_g)
    GradAcc acc;                                                           // deterministic mode: where the bias / seg_embed column sums go (common.h)
};

template <class T> __device__ __forceinline__ float ldT(const char* img, int pitch, int row, int d) {
    return to_f(*(const T*)(img + row * pitch + d * (int)sizeof(T)));
}

// ================================================================================================ forward
template <class T, int LP, int NW>
__global__ void __launch_bounds__(NW * 64) xl_attn_fwd_kernel(const T* __restrict__ qkv, const T* __restrict__ kr,
                                                              XlParams xp, T* __restrict__ vec, T* __restrict__ psave,
                                                              int L, int nh, DropKey drop) {
    drop.resolve();
    typedef AttnCfg<T> C;
    constexpr int PIT = C::ROWB + 16;
    constexpr int SPIT = LP * (int)sizeof(T) + 16;
    constexpr int NT = LP / 16, DSL = 64 / C::SLAB, LSL = LP / C::SLAB;
    // One block = one (batch, head) and NW consecutive 16-row query strips, one per wave (gridDim.y = NT / NW strip groups: one for
    // L <= 64, more for L <= 128 where all strips of a head do not fit next to the images).  Staged: the QR = NW * 16 query rows of
    // the group, all keys / values, and the window of relative positions those rows can reach -- row i reads raw[i][L - i + j], so
    // a group needs L + QR - 1 positions (RW tiles, block window starting at tile pt_lo) and a single strip L + 15 (RWT tiles
    // starting at pt0).  The probability strip re-uses the raw strip's LDS (written after the last raw read of the wave, which
    // executes in lock-step).
    static_assert(NT % NW == 0, "strip groups of NW strips");
    constexpr int QR = NW * 16;
    constexpr int RW = NT + NW + 1 < 2 * NT ? NT + NW + 1 : 2 * NT;      // position tiles staged per block
    constexpr int RR = RW * 16;
    constexpr int RWT = 2 * NT < NT + 2 ? 2 * NT : NT + 2;               // position tiles of one strip's window
    // LDS budget (round 4): at L <= 64 the block held 75 KB -- two blocks per CU, 512 slots for the 576 (batch, head) blocks of the
    // benchmarked shape: TWO rounds.  Three changes bring it to 51.8 KB = three blocks per CU, one round: the query rows are not
    // staged (a wave's 16 rows are only ever its own MFMA operand: loaded from global memory straight into fragment registers);
    // the raw shifted-score strip is kept in the activation dtype (bf16 mode: (q + r_r_bias) . kr rounded once more, the same
    // 2^-9 the operands already carry; fp32 mode: unchanged); the seg_embed image keeps its two rows (the 14 rows of padding a
    // 16-row MFMA operand reads fall into the strips behind it: their products land in accumulator rows nobody looks at).  The
    // probabilities stay in the accumulator registers (AccOp, as in attention.hip): no probability strip, no barrier in front of P.V.
    constexpr int RPIT = RWT * 16 * (int)sizeof(T) + 16;            // raw strip pitch
    __shared__ __attribute__((aligned(16))) char smem[(2 * LP + RR + 2) * PIT + NW * 16 * RPIT + (LP + RR + 4 + 2 * LP) * 4];
    char* Ki = smem;
    char* Vi = Ki + LP * PIT;
    char* Ri = Vi + LP * PIT;                          // KR image: positions [pt_lo * 16, pt_lo * 16 + RR)
    char* Si = Ri + RR * PIT;                          // seg_embed image, rows 0 and 1 (see above)
    char* raws = Si + 2 * PIT;
    float* cK = (float*)(raws + NW * 16 * RPIT);
    float* cR = cK + LP;
    float* cS = cR + RR;
    int* segv = (int*)(cS + 4);
    int* padf = segv + LP;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x / nh, h = blockIdx.x % nh;
    const int H = nh * 64;
    const size_t ld = (size_t)3 * H;
    const T* base = qkv + (size_t)b * L * ld + h * 64;
    const int rb = blockIdx.y * QR;                    // first query row of the group
    int pt_lo = (L - rb - QR + 1) >> 4;                // first position tile any row of the group can touch
    pt_lo = pt_lo < 0 ? 0 : (pt_lo > 2 * NT - RW ? 2 * NT - RW : pt_lo);
    {
        const int qv = L - rb < 0 ? 0 : (L - rb > QR ? QR : L - rb), rvv = 2 * L - pt_lo * 16 < 0 ? 0 : (2 * L - pt_lo * 16 > RR ? RR : 2 * L - pt_lo * 16);
        const T* rsrc = kr + ((size_t)b * 2 * L + (size_t)pt_lo * 16) * H + h * 64;
        (void)qv;
        if constexpr (LP <= 64) {
            char* const img[3] = {Ki, Vi, Ri};
            const T* const src[3] = {base + H, base + 2 * H, rsrc};
            const size_t lds[3] = {ld, ld, (size_t)H};
            const int ra[3] = {LP, LP, RR}, rv[3] = {L, L, rvv};
            stage_heads_var<T, NW * 64, 3, (RR > LP ? RR : LP)>(img, PIT, src, lds, ra, rv);       // every load in flight before the first LDS store
        } else {
            stage_rows<T, NW * 64>(Ki, PIT, base + H, ld, LP, L);
            stage_rows<T, NW * 64>(Vi, PIT, base + 2 * H, ld, LP, L);
            stage_rows<T, NW * 64>(Ri, PIT, rsrc, (size_t)H, RR, rvv);
        }
    }
    // this wave's query rows as MFMA operand fragments (rows >= L: zeros)
    typename Frag<T>::type qf[DSL];
    {
        const int qi = rb + wave * 16 + (lane & 15);
#pragma unroll
        for (int sl = 0; sl < DSL; ++sl) {
            qf[sl] = typename Frag<T>::type{};
            if (qi < L) qf[sl] = *(const typename Frag<T>::type*)(base + (size_t)qi * ld + sl * C::SLAB + (lane >> 4) * C::EPV);
        }
    }
    for (int t = threadIdx.x; t < 2 * 64; t += NW * 64) {
        const int row = t >> 6, d = t & 63;
        *(T*)(Si + row * PIT + d * (int)sizeof(T)) = from_f<T>(xp.seg_embed[((size_t)row * nh + h) * 64 + d]);
    }
    for (int j = threadIdx.x; j < LP; j += NW * 64) {
        segv[j] = j < L ? (int)xp.seg[(size_t)b * L + j] : -1;
        padf[j] = (j < L && xp.mask[(size_t)b * L + j] == 0) ? 1 : 0;
    }
    __syncthreads();
    const float* rwb = xp.r_w_bias + h * 64;
    const float* rrb = xp.r_r_bias + h * 64;
    const float* rsb = xp.r_s_bias + h * 64;
    for (int t = threadIdx.x; t < LP + RR + 2; t += NW * 64) {
        float s = 0.f;
        if (t < LP) { for (int d = 0; d < 64; ++d) s += rwb[d] * ldT<T>(Ki, PIT, t, d); cK[t] = s; }
        else if (t < LP + RR) { for (int d = 0; d < 64; ++d) s += rrb[d] * ldT<T>(Ri, PIT, t - LP, d); cR[t - LP] = s; }
        else { for (int d = 0; d < 64; ++d) s += rsb[d] * ldT<T>(Si, PIT, t - LP - RR, d); cS[t - LP - RR] = s; }
    }
    __syncthreads();

    char* raw = raws + wave * 16 * RPIT;
    const float scale = 0.125f;
    {
        const int strip = blockIdx.y * NW + wave;       // global strip of this wave; its query rows are rows wave * 16 .. of Qi
        const bool active = true;
        f32x4 ac[NT];
        float e0 = 0.f, e1 = 0.f;
        int pt0 = (L - strip * 16 - 15) >> 4;          // first position tile any row of this strip can touch
        pt0 = pt0 < 0 ? 0 : (pt0 > 2 * NT - RWT ? 2 * NT - RWT : pt0);
        pt0 = pt0 > pt_lo + RW - RWT ? pt_lo + RW - RWT : pt0;      // ... inside the block's staged window
        if (active) {
#pragma unroll
            for (int jt = 0; jt < NT; ++jt) {
                ac[jt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int sl = 0; sl < DSL; ++sl)
                    mma16(ac[jt], frag_nat<T>(Ki, PIT, jt * 16 + (lane & 15), sl, lane), qf[sl]);
            }
#pragma unroll
            for (int q = 0; q < RWT; ++q) {
                const int pt = pt0 + q;
                f32x4 rw = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int sl = 0; sl < DSL; ++sl)
                    mma16(rw, frag_nat<T>(Ri, PIT, (pt - pt_lo) * 16 + (lane & 15), sl, lane), qf[sl]);
                const int p0 = pt * 16 + (lane >> 4) * 4;
                rw += *(const f32x4*)(cR + p0 - pt_lo * 16);
                store4((T*)(raw + (lane & 15) * RPIT) + (p0 - pt0 * 16), rw);          // raw[i][p] = (q_i + r_r_bias) . kr_p
            }
            f32x4 ev = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int sl = 0; sl < DSL; ++sl)
                mma16(ev, frag_nat<T>(Si, PIT, lane & 15, sl, lane), qf[sl]);
            e0 = __shfl(ev[0] + cS[0], lane & 15, 64);     // lanes 0..15 hold E[i][s = 0, 1]
            e1 = __shfl(ev[1] + cS[1], lane & 15, 64);
        }
        __syncthreads();
        if (active) {
            const int i = strip * 16 + (lane & 15);
            const int si = segv[i];
            float mx = -3.0e38f;
#pragma unroll
            for (int jt = 0; jt < NT; ++jt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int j = jt * 16 + (lane >> 4) * 4 + r;
                    int p = L - i + j - pt0 * 16;          // inside the window for every real (i < L, j < L) pair
                    p = p < 0 ? 0 : (p > RWT * 16 - 1 ? RWT * 16 - 1 : p);
                    const float bd = to_f(*(const T*)(raw + (lane & 15) * RPIT + p * (int)sizeof(T)));
                    float s = (ac[jt][r] + cK[j] + bd + (si == segv[j] ? e0 : e1)) * scale;
                    if (j >= L) s = kPadNeg;
                    else if ((i != j || xp.gstream) && (padf[j] || (xp.perm != nullptr && i < L && xp.perm[((size_t)b * L + i) * L + j] != 0))) s -= kXlMask;
                    ac[jt][r] = s;
                    mx = fmaxf(mx, s);
                }
            }
            mx = quad_max(mx);
            float sum = 0.f;
#pragma unroll
            for (int jt = 0; jt < NT; ++jt)
#pragma unroll
                for (int r = 0; r < 4; ++r) { ac[jt][r] = __expf(ac[jt][r] - mx); sum += ac[jt][r]; }
            const float inv = 1.0f / quad_sum(sum);
            const uint32_t rowidx = ((uint32_t)blockIdx.x * L + (uint32_t)i) * L;
#pragma unroll
            for (int jt = 0; jt < NT; ++jt) {
                const int j = jt * 16 + (lane >> 4) * 4;
                f32x4 p = ac[jt] * inv;
                // saved probabilities: rows padded to LP columns -> one aligned 4-element store (columns >= L hold exact zeros)
                if (psave != nullptr && i < L) store4(psave + ((size_t)blockIdx.x * LP + i) * LP + j, p);
#pragma unroll
                for (int r = 0; r < 4; ++r) p[r] *= drop_mult(drop, rowidx + j + r);
                ac[jt] = p;                      // the dropped probabilities: operand of P.V below (AccOp)
            }
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int sl = 0; sl < LSL; ++sl)
                    mma16(o, AccOp<T>::kmaj(Vi, PIT, sl, dt * 16 + (lane & 15), lane), AccOp<T>::make(&ac[sl * AccOp<T>::TILES]));
                const int i = strip * 16 + (lane & 15);
                if (xp.head_scale) o = o * xp.head_scale[h];      // attn_prob * head_mask, applied to the head's output (same product)
                if (i < L) store4(vec + ((size_t)b * L + i) * H + h * 64 + dt * 16 + (lane >> 4) * 4, o);
            }
        }
        __syncthreads();
    }
}

// block-level column-sum flush of N [4][4] per-lane tiles at once: per-lane partials (own row only) -> 16-row DPP reduction ->
// waves summed in LDS -> ONE atomic per column.  `scratch` holds N * NW * 64 floats and must be free (call after a barrier).
// One pass for all five bias / segment-embedding gradients: flushed one tile at a time, each flush ended in a barrier that
// waited for its atomics' round trip to L2.
template <int NW, int N>
__device__ __forceinline__ void flush_colsums(f32x4 (* const (&c4)[N])[4], float* const (&dst64)[N], float* scratch, int lane, int wave, const GradAcc& acc) {
#pragma unroll
    for (int n = 0; n < N; ++n)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float s = row16_sum_to_lane15((*c4[n])[dt][r]);
                if ((lane & 15) == 15) scratch[(n * NW + wave) * 64 + dt * 16 + (lane >> 4) * 4 + r] = s;
            }
    __syncthreads();
    for (int j = threadIdx.x; j < N * 64; j += NW * 64) {
        const int n = j >> 6, col = j & 63;
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) t += scratch[(n * NW + w) * 64 + col];
        grad_add(acc, dst64[n] + col, t);
    }
}

// ================================================================================================ backward, query side
template <class T, int LP, int NW>
__device__ __forceinline__ void xl_attn_bwd_q_body(const T* __restrict__ qkv, const T* __restrict__ kr, XlParams xp,
                                                   const T* __restrict__ psave, const T* __restrict__ dvec,
                                                   T* __restrict__ gsave, T* __restrict__ dqkv, float* d_rwb,
                                                   float* d_rrb, float* d_rsb, float* d_seg, int L, int nh,
                                                   DropKey drop) {
    drop.resolve();
    typedef AttnCfg<T> C;
    constexpr int PIT = C::ROWB + 16;
    constexpr int SPIT = LP * (int)sizeof(T) + 16;
    constexpr int NT = LP / 16, DSL = 64 / C::SLAB, LSL = LP / C::SLAB;
    // strip groups and position windows as in xl_attn_fwd_kernel: NW query strips per block (gridDim.y = NT / NW groups), the
    // block's RW position tiles of KR staged from tile pt_lo on, a strip's shifted score gradients kept for its RWT-tile window only
    static_assert(NT % NW == 0, "strip groups of NW strips");
    constexpr int QR = NW * 16;
    constexpr int RW = NT + NW + 1 < 2 * NT ? NT + NW + 1 : 2 * NT;
    constexpr int RR = RW * 16;
    constexpr int RWT = 2 * NT < NT + 2 ? 2 * NT : NT + 2;
    constexpr int GPIT = RWT * 16 * (int)sizeof(T) + 16;     // shifted-G strip pitch (the strip's position window)
    constexpr int RSL = RWT * 16 / C::SLAB;
    // Q is not staged: it is only read once per row at the end (q_i + r_s_bias for the segment-embedding gradient), straight from
    // HBM/L2.  Round 4 (LDS 69.9 -> 51.5 KB at L <= 64: three blocks per CU, the 576 blocks of the benchmarked shape in ONE round
    // instead of two): the dvec rows of a wave are only its own MFMA operand -> fragment registers loaded from global memory, no
    // image; the score gradients G stay in the accumulator registers as the operand of G . K (AccOp) -> no natural-order strip,
    // only the position-shifted one.
    (void)SPIT;
    __shared__ __attribute__((aligned(16))) char smem[(2 * LP + RR) * PIT + NW * 16 * GPIT + (192 + 2 * LP) * 4];
    char* Ki = smem;
    char* Vi = Ki + LP * PIT;
    char* Ri = Vi + LP * PIT;                          // KR image: positions [pt_lo * 16, pt_lo * 16 + RR)
    char* sstr = Ri + RR * PIT;
    float* sef = (float*)(sstr + NW * 16 * GPIT);     // se0[64] | se1[64] | rsb[64]
    int* segv = (int*)(sef + 192);
    int* padf = segv + LP;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x / nh, h = blockIdx.x % nh;
    const int H = nh * 64;
    const size_t ld = (size_t)3 * H;
    const T* base = qkv + (size_t)b * L * ld + h * 64;
    const int rb = blockIdx.y * QR;
    int pt_lo = (L - rb - QR + 1) >> 4;
    pt_lo = pt_lo < 0 ? 0 : (pt_lo > 2 * NT - RW ? 2 * NT - RW : pt_lo);
    {
        const int qv = L - rb < 0 ? 0 : (L - rb > QR ? QR : L - rb), rvv = 2 * L - pt_lo * 16 < 0 ? 0 : (2 * L - pt_lo * 16 > RR ? RR : 2 * L - pt_lo * 16);
        const T* osrc = dvec + ((size_t)b * L + rb) * H + h * 64;
        const T* rsrc = kr + ((size_t)b * 2 * L + (size_t)pt_lo * 16) * H + h * 64;
        (void)qv; (void)osrc;
        if constexpr (LP <= 64) {
            char* const img[3] = {Ki, Vi, Ri};
            const T* const src[3] = {base + H, base + 2 * H, rsrc};
            const size_t lds[3] = {ld, ld, (size_t)H};
            const int ra[3] = {LP, LP, RR}, rv[3] = {L, L, rvv};
            stage_heads_var<T, NW * 64, 3, (RR > LP ? RR : LP)>(img, PIT, src, lds, ra, rv);
        } else {
            stage_rows<T, NW * 64>(Ki, PIT, base + H, ld, LP, L);
            stage_rows<T, NW * 64>(Vi, PIT, base + 2 * H, ld, LP, L);
            stage_rows<T, NW * 64>(Ri, PIT, rsrc, (size_t)H, RR, rvv);
        }
    }
    // this wave's dvec rows as MFMA operand fragments (rows >= L: zeros)
    typename Frag<T>::type of[DSL];
    {
        const int oi = rb + wave * 16 + (lane & 15);
#pragma unroll
        for (int sl = 0; sl < DSL; ++sl) {
            of[sl] = typename Frag<T>::type{};
            if (oi < L) of[sl] = *(const typename Frag<T>::type*)(dvec + ((size_t)b * L + oi) * H + h * 64 + sl * C::SLAB + (lane >> 4) * C::EPV);
        }
    }
    for (int t = threadIdx.x; t < 192; t += NW * 64)
        sef[t] = t < 128 ? xp.seg_embed[((size_t)(t >> 6) * nh + h) * 64 + (t & 63)] : xp.r_s_bias[h * 64 + (t & 63)];
    for (int j = threadIdx.x; j < LP; j += NW * 64) {
        segv[j] = j < L ? (int)xp.seg[(size_t)b * L + j] : -1;
        padf[j] = 0;
    }
    __syncthreads();
    const float hs = xp.head_scale ? xp.head_scale[h] : 1.0f;      // head_mask: every gradient of this head is linear in its dvec

    f32x4 cw[4], cr[4], cs[4], d0[4], d1[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) cw[dt] = cr[dt] = cs[dt] = d0[dt] = d1[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    char* Ss = sstr + wave * 16 * GPIT;
    const float scale = 0.125f;
    T* dq_base = dqkv + (size_t)b * L * ld + h * 64;

    {
        const int strip = blockIdx.y * NW + wave;
        const bool active = true;
        const int i = strip * 16 + (lane & 15);
        int pt0 = (L - strip * 16 - 15) >> 4;          // the strip's position window (as in the forward)
        pt0 = pt0 < 0 ? 0 : (pt0 > 2 * NT - RWT ? 2 * NT - RWT : pt0);
        pt0 = pt0 > pt_lo + RW - RWT ? pt_lo + RW - RWT : pt0;
        float g0 = 0.f, g1 = 0.f;
        f32x4 pv[NT];                          // P, then G: stays in registers as the operand of G . K
        if (active) {
            // zero this wave's shifted strip, then G in registers and G shifted to position space
            for (int t = lane; t < 16 * GPIT / 16; t += 64) *(u32x4*)(Ss + t * 16) = u32x4{0u, 0u, 0u, 0u};
            f32x4 dp[NT];
#pragma unroll
            for (int jt = 0; jt < NT; ++jt) {
                dp[jt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int sl = 0; sl < DSL; ++sl)
                    mma16(dp[jt], frag_nat<T>(Vi, PIT, jt * 16 + (lane & 15), sl, lane), of[sl]);
                dp[jt] = dp[jt] * hs;
            }
            const uint32_t rowidx = ((uint32_t)blockIdx.x * L + (uint32_t)i) * L;
            float dsum = 0.f;
            const size_t prow = ((size_t)blockIdx.x * LP + (i < LP ? i : 0)) * LP;      // storage row (padded to LP columns)
#pragma unroll
            for (int jt = 0; jt < NT; ++jt) {
                const f32x4 pq = i < L ? load4(psave + prow + jt * 16 + (lane >> 4) * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int j = jt * 16 + (lane >> 4) * 4 + r;
                    const bool ok = i < L && j < L;
                    pv[jt][r] = ok ? pq[r] : 0.f;
                    dp[jt][r] *= ok ? drop_mult(drop, rowidx + j) : 0.f;
                    dsum += dp[jt][r] * pv[jt][r];
                }
            }
            const float D = quad_sum(dsum);
            const int si = segv[i < LP ? i : 0];
#pragma unroll
            for (int jt = 0; jt < NT; ++jt) {
                const int j0 = jt * 16 + (lane >> 4) * 4;
                f32x4 g = pv[jt] * (dp[jt] - D) * scale;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int j = j0 + r;
                    if (i < L && j < L) {
                        if (si == segv[j]) g0 += g[r]; else g1 += g[r];
                        *(T*)(Ss + (lane & 15) * GPIT + (L - i + j - pt0 * 16) * (int)sizeof(T)) = from_f<T>(g[r]);
                    } else g[r] = 0.f;
                }
                if (i < L) store4(gsave + prow + j0, g);          // same padded layout as psave
                pv[jt] = g;                                          // G stays in registers: the operand of G . K below (AccOp)
            }
            g0 = quad_sum(g0);
            g1 = quad_sum(g1);
        }
        __syncthreads();
        if (active) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                f32x4 oa = {0.f, 0.f, 0.f, 0.f}, ob = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int sl = 0; sl < LSL; ++sl)
                    mma16(oa, AccOp<T>::kmaj(Ki, PIT, sl, dt * 16 + (lane & 15), lane), AccOp<T>::make(&pv[sl * AccOp<T>::TILES]));
#pragma unroll
                for (int sl = 0; sl < RSL; ++sl)
                    mma16(ob, frag_kmaj(Ri, PIT, (pt0 - pt_lo) * 16 + sl * C::SLAB + (lane >> 4) * C::EPV, dt * 16 + (lane & 15), T()),
                          frag_nat<T>(Ss, GPIT, lane & 15, sl, lane));
                const int d = dt * 16 + (lane >> 4) * 4;
                const f32x4 s0v = *(const f32x4*)(sef + d), s1v = *(const f32x4*)(sef + 64 + d), rs = *(const f32x4*)(sef + 128 + d);
                const f32x4 oe = g0 * s0v + g1 * s1v;
                if (i < L) {
                    store4(dq_base + (size_t)i * ld + d, oa + ob + oe);
                    cw[dt] += oa; cr[dt] += ob; cs[dt] += oe;
                    const f32x4 qv = load4(base + (size_t)i * ld + d) + rs;          // q_i + r_s_bias
                    d0[dt] += g0 * qv; d1[dt] += g1 * qv;
                }
            }
        }
        __syncthreads();
    }
    {
        // the strip buffers are free after the last strip barrier: 5 * NW * 64 floats fit (NW * 16 * (SPIT + GPIT) bytes)
        static_assert(NW * 16 * GPIT >= 5 * NW * 64 * 4, "the shifted strips hold the five column-sum tiles");
        f32x4 (* const tiles[5])[4] = {&cw, &cr, &cs, &d0, &d1};
        float* const dst[5] = {d_rwb + h * 64, d_rrb + h * 64, d_rsb + h * 64, d_seg + (size_t)h * 64, d_seg + ((size_t)nh + h) * 64};
        flush_colsums<NW, 5>(tiles, dst, (float*)sstr, lane, wave, xp.acc);
    }
}
template <class T, int LP, int NW>
__global__ void __launch_bounds__(NW * 64) xl_attn_bwd_q_kernel(const T* __restrict__ qkv, const T* __restrict__ kr, XlParams xp,
                                                                const T* __restrict__ psave, const T* __restrict__ dvec,
                                                                T* __restrict__ gsave, T* __restrict__ dqkv, float* d_rwb,
                                                                float* d_rrb, float* d_rsb, float* d_seg, int L, int nh,
                                                                DropKey drop) {
    xl_attn_bwd_q_body<T, LP, NW>(qkv, kr, xp, psave, dvec, gsave, dqkv, d_rwb, d_rrb, d_rsb, d_seg, L, nh, drop);
}
// ... with AdamW riders (kernels.h AdamRide) behind its `nblk` (batch, head) workgroups: 576 blocks in 768 slots at L = 50 (one strip group:
// gridDim.y == 1), a latency-bound kernel -- as attention.hip's attn_bwd_ride_kernel
template <class T, int LP, int NW>
__global__ void __launch_bounds__(NW * 64) xl_attn_bwd_q_ride_kernel(const T* __restrict__ qkv, const T* __restrict__ kr, XlParams xp,
                                                                     const T* __restrict__ psave, const T* __restrict__ dvec,
                                                                     T* __restrict__ gsave, T* __restrict__ dqkv, float* d_rwb,
                                                                     float* d_rrb, float* d_rsb, float* d_seg, int L, int nh,
                                                                     DropKey drop, const AdamRide ride, int nblk) {
    if ((int)blockIdx.x >= nblk) {
        adam_ride_block<NW * 64, 2>(ride, (int)blockIdx.x - nblk);
        return;
    }
    xl_attn_bwd_q_body<T, LP, NW>(qkv, kr, xp, psave, dvec, gsave, dqkv, d_rwb, d_rrb, d_rsb, d_seg, L, nh, drop);
}


// ================================================================================================ backward, key / position side
template <class T, int LP, int NW>
__global__ void __launch_bounds__(NW * 64) xl_attn_bwd_kv_kernel(const T* __restrict__ qkv, XlParams xp, const T* __restrict__ psave,
                                                                 const T* __restrict__ gsave, const T* __restrict__ dvec,
                                                                 T* __restrict__ dqkv, T* __restrict__ dkr, int L, int nh,
                                                                 DropKey drop) {
    drop.resolve();
    typedef AttnCfg<T> C;
    constexpr int PIT = C::ROWB + 16;
    constexpr int SPIT = LP * (int)sizeof(T) + 16;
    constexpr int NT = LP / 16, LSL = LP / C::SLAB;
    __shared__ __attribute__((aligned(16))) char smem[2 * LP * PIT + NW * 16 * SPIT + 128 * 4];
    char* Qi = smem;
    char* Oi = Qi + LP * PIT;
    char* strips = Oi + LP * PIT;
    float* bia = (float*)(strips + NW * 16 * SPIT);     // r_w_bias[64] | r_r_bias[64]

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x / nh, h = blockIdx.x % nh;
    const int H = nh * 64;
    const size_t ld = (size_t)3 * H;
    {
        char* const img[2] = {Qi, Oi};
        const T* const src[2] = {qkv + (size_t)b * L * ld + h * 64, dvec + (size_t)b * L * H + h * 64};
        const size_t lds[2] = {ld, (size_t)H};
        stage_heads<T, LP, NW * 64, 2>(img, PIT, src, lds, L);
    }
    for (int t = threadIdx.x; t < 128; t += NW * 64) bia[t] = t < 64 ? xp.r_w_bias[h * 64 + t] : xp.r_r_bias[h * 64 + t - 64];
    __syncthreads();
    if (xp.head_scale) {
        scale_image<T, LP, NW * 64>(Oi, PIT, xp.head_scale[h]);
        __syncthreads();
    }
    char* St = strips + wave * 16 * SPIT;
    const size_t pbase = (size_t)blockIdx.x * LP * LP;       // psave / gsave rows are padded to LP columns
    const size_t dbase = (size_t)blockIdx.x * L * L;         // dropout element index space (unpadded)
    T* dq_base = dqkv + (size_t)b * L * ld + h * 64;

    // ---- key strips: dv, dk
    for (int s0 = 0; s0 < NT; s0 += NW) {
        const int strip = s0 + wave;
        const bool active = strip < NT;
        const int j = strip * 16 + (lane & 15);
        f32x4 gt[NT];
        float csum = 0.f;
        if (active) {
#pragma unroll
            for (int it = 0; it < NT; ++it) {
                const int i0 = it * 16 + (lane >> 4) * 4;
                f32x4 pd;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = i0 + r;
                    const bool ok = i < L && j < L;
                    const size_t idx = pbase + (size_t)i * LP + j;
                    pd[r] = ok ? to_f(psave[idx]) * drop_mult(drop, (uint32_t)(dbase + (size_t)i * L + j)) : 0.f;
                    gt[it][r] = ok ? to_f(gsave[idx]) : 0.f;
                    csum += gt[it][r];
                }
                store4((T*)(St + (lane & 15) * SPIT) + i0, pd);
            }
            csum = quad_sum(csum);
        }
        __syncthreads();
        if (active) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int sl = 0; sl < LSL; ++sl)
                    mma16(o, frag_kmaj(Oi, PIT, sl * C::SLAB + (lane >> 4) * C::EPV, dt * 16 + (lane & 15), T()),
                          frag_nat<T>(St, SPIT, lane & 15, sl, lane));
                if (j < L) store4(dq_base + (size_t)j * ld + 2 * H + dt * 16 + (lane >> 4) * 4, o);
            }
        }
        __syncthreads();
        if (active) {
#pragma unroll
            for (int it = 0; it < NT; ++it) store4((T*)(St + (lane & 15) * SPIT) + it * 16 + (lane >> 4) * 4, gt[it]);
        }
        __syncthreads();
        if (active) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int sl = 0; sl < LSL; ++sl)
                    mma16(o, frag_kmaj(Qi, PIT, sl * C::SLAB + (lane >> 4) * C::EPV, dt * 16 + (lane & 15), T()),
                          frag_nat<T>(St, SPIT, lane & 15, sl, lane));
                const int d = dt * 16 + (lane >> 4) * 4;
                o += csum * *(const f32x4*)(bia + d);                 // + (sum_i G[i,j]) * r_w_bias
                if (j < L) store4(dq_base + (size_t)j * ld + H + d, o);
            }
        }
        __syncthreads();
    }
    // ---- position strips: dkr[p] = sum_i G[i, p - L + i] (q_i + r_r_bias)
    for (int s0 = 0; s0 < 2 * NT; s0 += NW) {
        const int strip = s0 + wave;
        const bool active = strip < 2 * NT;
        const int p = strip * 16 + (lane & 15);
        float csum = 0.f;
        if (active) {
#pragma unroll
            for (int it = 0; it < NT; ++it) {
                const int i0 = it * 16 + (lane >> 4) * 4;
                f32x4 g;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = i0 + r, j = p - L + i;
                    g[r] = (i < L && j >= 0 && j < L) ? to_f(gsave[pbase + (size_t)i * LP + j]) : 0.f;
                    csum += g[r];
                }
                store4((T*)(St + (lane & 15) * SPIT) + i0, g);
            }
            csum = quad_sum(csum);
        }
        __syncthreads();
        if (active) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int sl = 0; sl < LSL; ++sl)
                    mma16(o, frag_kmaj(Qi, PIT, sl * C::SLAB + (lane >> 4) * C::EPV, dt * 16 + (lane & 15), T()),
                          frag_nat<T>(St, SPIT, lane & 15, sl, lane));
                const int d = dt * 16 + (lane >> 4) * 4;
                o += csum * *(const f32x4*)(bia + 64 + d);
                if (p < 2 * L) store4(dkr + ((size_t)b * 2 * L + p) * H + h * 64 + d, o);
            }
        }
        __syncthreads();
    }
}

// ---- the same for L <= 64 in bf16 (round 4): P (with this step's dropout applied) and G of the head are staged ONCE into LDS with
// coalesced 16-byte loads, and every product reads them through transposed fragment reads -- dv = Pd^T dO and dk = G^T (q + r_w_bias)
// take BOTH operands k-major (k = the query index), so there are no per-element global loads (the kernel above fetches its tiles with
// 2-byte loads in transposed order: 48 dependent round trips per wave), no strips and no barriers for the key side; the position
// side gathers its diagonals out of the LDS image instead of out of HBM.  46.6 KB of LDS: three blocks per CU.
template <class T, int LP, int NW>
__device__ __forceinline__ void xl_attn_bwd_kv2_body(const T* __restrict__ qkv, XlParams xp, const T* __restrict__ psave,
                                                     const T* __restrict__ gsave, const T* __restrict__ dvec,
                                                     T* __restrict__ dqkv, T* __restrict__ dkr, int L, int nh,
                                                     DropKey drop) {
    drop.resolve();
    typedef AttnCfg<T> C;
    constexpr int PIT = C::ROWB + 16;
    constexpr int SPIT = LP * (int)sizeof(T) + 16;
    constexpr int NT = LP / 16, LSL = LP / C::SLAB;
    constexpr int CPR = LP * (int)sizeof(T) / 16;           // 16-byte chunks per P / G row
    __shared__ __attribute__((aligned(16))) char smem[2 * LP * PIT + 2 * LP * SPIT + NW * 16 * SPIT + 128 * 4];
    char* Qi = smem;
    char* Oi = Qi + LP * PIT;
    char* Pi = Oi + LP * PIT;                           // Pd[i][j]: saved probabilities x this step's dropout multipliers
    char* Gi = Pi + LP * SPIT;                          // G[i][j]
    char* strips = Gi + LP * SPIT;
    float* bia = (float*)(strips + NW * 16 * SPIT);     // r_w_bias[64] | r_r_bias[64]

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x / nh, h = blockIdx.x % nh;
    const int H = nh * 64;
    const size_t ld = (size_t)3 * H;
    const size_t pbase = (size_t)blockIdx.x * LP * LP;       // psave / gsave rows are padded to LP columns
    const size_t dbase = (size_t)blockIdx.x * L * L;         // dropout element index space (unpadded)
    {
        // P and G rows first (their loads are in flight under the staging of Q and dvec below); rows >= L were never written
        constexpr int IT = (LP * CPR + NW * 64 - 1) / (NW * 64);
        u32x4 pv[IT], gv[IT];
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int id = threadIdx.x + it * NW * 64, row = id / CPR, c = id % CPR;
            pv[it] = gv[it] = u32x4{0u, 0u, 0u, 0u};
            if (id < LP * CPR && row < L) {
                pv[it] = *(const u32x4*)((const char*)(psave + pbase + (size_t)row * LP) + c * 16);
                gv[it] = *(const u32x4*)((const char*)(gsave + pbase + (size_t)row * LP) + c * 16);
            }
        }
        char* const img[2] = {Qi, Oi};
        const T* const src[2] = {qkv + (size_t)b * L * ld + h * 64, dvec + (size_t)b * L * H + h * 64};
        const size_t lds[2] = {ld, (size_t)H};
        stage_heads<T, LP, NW * 64, 2>(img, PIT, src, lds, L);
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int id = threadIdx.x + it * NW * 64, row = id / CPR, c = id % CPR;
            if (id >= LP * CPR) continue;
            union { u32x4 u; T e[C::EPV]; } x;
            x.u = pv[it];
            if (row < L) {
#pragma unroll
                for (int q = 0; q < C::EPV; ++q) {
                    const int j = c * C::EPV + q;
                    x.e[q] = j < L ? from_f<T>(to_f(x.e[q]) * drop_mult(drop, (uint32_t)(dbase + (size_t)row * L + j))) : from_f<T>(0.f);
                }
            }
            *(u32x4*)(Pi + row * SPIT + c * 16) = x.u;
            *(u32x4*)(Gi + row * SPIT + c * 16) = gv[it];
        }
    }
    for (int t = threadIdx.x; t < 128; t += NW * 64) bia[t] = t < 64 ? xp.r_w_bias[h * 64 + t] : xp.r_r_bias[h * 64 + t - 64];
    __syncthreads();
    if (xp.head_scale) {
        scale_image<T, LP, NW * 64>(Oi, PIT, xp.head_scale[h]);
        __syncthreads();
    }
    T* dq_base = dqkv + (size_t)b * L * ld + h * 64;

    // ---- key strips: dv, dk (a wave owns whole strips; nothing is written to LDS)
    for (int strip = wave; strip < NT; strip += NW) {
        const int j = strip * 16 + (lane & 15);
        float csum = 0.f;
#pragma unroll
        for (int m = 0; m < LP / 4; ++m) csum += to_f(*(const T*)(Gi + ((lane >> 4) + 4 * m) * SPIT + j * (int)sizeof(T)));
        csum = quad_sum(csum);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            f32x4 ov = {0.f, 0.f, 0.f, 0.f}, ok = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int sl = 0; sl < LSL; ++sl) {
                const int k0 = sl * C::SLAB + (lane >> 4) * C::EPV;
                mma16(ov, frag_kmaj(Oi, PIT, k0, dt * 16 + (lane & 15), T()), frag_kmaj(Pi, SPIT, k0, j, T()));
                mma16(ok, frag_kmaj(Qi, PIT, k0, dt * 16 + (lane & 15), T()), frag_kmaj(Gi, SPIT, k0, j, T()));
            }
            const int d = dt * 16 + (lane >> 4) * 4;
            ok += csum * *(const f32x4*)(bia + d);                // + (sum_i G[i,j]) * r_w_bias
            if (j < L) {
                store4(dq_base + (size_t)j * ld + 2 * H + d, ov);
                store4(dq_base + (size_t)j * ld + H + d, ok);
            }
        }
    }
    // ---- position strips: dkr[p] = sum_i G[i, p - L + i] (q_i + r_r_bias); the diagonals are gathered out of the G image into the
    // wave's own strip (written and read by this wave only: LDS operations of a wave execute in order, no block barrier)
    char* St = strips + wave * 16 * SPIT;
    for (int strip = wave; strip < 2 * NT; strip += NW) {
        const int p = strip * 16 + (lane & 15);
        float csum = 0.f;
#pragma unroll
        for (int it = 0; it < NT; ++it) {
            const int i0 = it * 16 + (lane >> 4) * 4;
            f32x4 g;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = i0 + r, j = p - L + i;
                g[r] = (i < L && j >= 0 && j < L) ? to_f(*(const T*)(Gi + i * SPIT + j * (int)sizeof(T))) : 0.f;
                csum += g[r];
            }
            store4((T*)(St + (lane & 15) * SPIT) + i0, g);
        }
        csum = quad_sum(csum);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int sl = 0; sl < LSL; ++sl)
                mma16(o, frag_kmaj(Qi, PIT, sl * C::SLAB + (lane >> 4) * C::EPV, dt * 16 + (lane & 15), T()),
                      frag_nat<T>(St, SPIT, lane & 15, sl, lane));
            const int d = dt * 16 + (lane >> 4) * 4;
            o += csum * *(const f32x4*)(bia + 64 + d);
            if (p < 2 * L) store4(dkr + ((size_t)b * 2 * L + p) * H + h * 64 + d, o);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}
template <class T, int LP, int NW>
__global__ void __launch_bounds__(NW * 64) xl_attn_bwd_kv2_kernel(const T* __restrict__ qkv, XlParams xp, const T* __restrict__ psave,
                                                                  const T* __restrict__ gsave, const T* __restrict__ dvec,
                                                                  T* __restrict__ dqkv, T* __restrict__ dkr, int L, int nh,
                                                                  DropKey drop) {
    xl_attn_bwd_kv2_body<T, LP, NW>(qkv, xp, psave, gsave, dvec, dqkv, dkr, L, nh, drop);
}
template <class T, int LP, int NW>
__global__ void __launch_bounds__(NW * 64) xl_attn_bwd_kv2_ride_kernel(const T* __restrict__ qkv, XlParams xp, const T* __restrict__ psave,
                                                                       const T* __restrict__ gsave, const T* __restrict__ dvec,
                                                                       T* __restrict__ dqkv, T* __restrict__ dkr, int L, int nh,
                                                                       DropKey drop, const AdamRide ride, int nblk) {
    if ((int)blockIdx.x >= nblk) {
        adam_ride_block<NW * 64, 2>(ride, (int)blockIdx.x - nblk);
        return;
    }
    xl_attn_bwd_kv2_body<T, LP, NW>(qkv, xp, psave, gsave, dvec, dqkv, dkr, L, nh, drop);
}


// ================================================================================================ host
// LP = L rounded up to 32 / 64 / 128; NW_* = waves (query strips per block for the two query-side kernels, whose grid has
// LP / 16 / NW strip groups in y; plain wave count for the key-side kernel, which loops over its strips).  At LP = 128 the images of a
// whole head leave room for two strips (one in fp32 for the backward, whose strips also keep the shifted gradients).
#define XL_DISPATCH(KERNEL_CALL)                                                   \
    if (L < 1 || L > 128) return MB_ERR_SHAPE;                                     \
    {                                                                              \
        const int LPv = L <= 32 ? 32 : (L <= 64 ? 64 : 128);                       \
        if (dtype == DT_BF16) { typedef bf16 T;                                    \
            if (LPv == 32) { constexpr int LP = 32, NWF = 2, NWQ = 2, NWK = 2; KERNEL_CALL }          \
            else if (LPv == 64) { constexpr int LP = 64, NWF = 4, NWQ = 4, NWK = 4; KERNEL_CALL }     \
            else { constexpr int LP = 128, NWF = 2, NWQ = 2, NWK = 8; KERNEL_CALL } }                 \
        else if (dtype == DT_F32) { typedef float T;                               \
            if (LPv == 32) { constexpr int LP = 32, NWF = 2, NWQ = 2, NWK = 2; KERNEL_CALL }          \
            else if (LPv == 64) { constexpr int LP = 64, NWF = 4, NWQ = 4, NWK = 4; KERNEL_CALL }     \
            else { constexpr int LP = 128, NWF = 2, NWQ = 1, NWK = 8; KERNEL_CALL } }                 \
        else return MB_ERR_DTYPE;                                                  \
    }                                                                              \
    return (int)hipGetLastError();

int xlnet_attention_forward(int dtype, const void* qkv, const void* kr, const float* r_w_bias, const float* r_r_bias,
                            const float* r_s_bias, const float* seg_embed, const int64_t* seg, const int64_t* mask, void* vec,
                            void* psave, int B, int L, int nh, DropKey drop, hipStream_t st, const float* head_scale, const uint8_t* perm,
                            int gstream) {
    XlParams xp = {r_w_bias, r_r_bias, r_s_bias, seg_embed, seg, mask, head_scale, perm, gstream, GradAcc{nullptr, nullptr}};
    XL_DISPATCH({
        (void)NWQ; (void)NWK;
        hipLaunchKernelGGL((xl_attn_fwd_kernel<T, LP, NWF>), dim3(B * nh, LP / 16 / NWF), dim3(NWF * 64), 0, st, (const T*)qkv, (const T*)kr, xp,
                           (T*)vec, (T*)psave, L, nh, drop);
    })
}

// block slots the two backward launches of nblk (batch, head) workgroups leave free in their last round, when they can carry riders
// (bf16, L <= 64: 51 / 47 KB of LDS = three blocks per CU each); 0 = no riders for this shape
int xlnet_attention_backward_free_slots(int dtype, int L, int nblk, int cus) {
    if (dtype != DT_BF16 || L < 1 || L > 64) return 0;
    static int kv2 = -1;
    if (kv2 < 0) { const char* v = getenv("MB_XL_KV2"); kv2 = v ? atoi(v) : 1; }
    if (!kv2) return 0;
    const int slots = 3 * cus, rounds = (nblk + slots - 1) / slots;
    return rounds * slots - nblk;
}

int xlnet_attention_backward(int dtype, const void* qkv, const void* kr, const float* r_w_bias, const float* r_r_bias,
                             const float* r_s_bias, const float* seg_embed, const int64_t* seg, const int64_t* mask,
                             const void* psave, const void* dvec, void* gsave, void* dqkv, void* dkr, float* d_rwb,
                             float* d_rrb, float* d_rsb, float* d_seg, int B, int L, int nh, DropKey drop, hipStream_t st,
                             const float* head_scale, GradAcc acc, const AdamRide* ride_q, const AdamRide* ride_kv) {
    static int g_kv2 = -1;            // MB_XL_KV2=0: the key / position side of the backward with the round-3 kernel at L <= 64 too (A/B)
    if (g_kv2 < 0) { const char* v = getenv("MB_XL_KV2"); g_kv2 = v ? atoi(v) : 1; }
    XlParams xp = {r_w_bias, r_r_bias, r_s_bias, seg_embed, seg, mask, head_scale, nullptr, 0, acc};
    XL_DISPATCH({
        (void)NWF;
        bool q_done = false;
        if constexpr (sizeof(T) == 2 && LP <= 64 && LP / 16 / NWQ == 1) {
            if (ride_q != nullptr && ride_q->blocks > 0 && ride_q->n4 > 0) {
                gemm_log_ride(*ride_q);
                hipLaunchKernelGGL((xl_attn_bwd_q_ride_kernel<T, LP, NWQ>), dim3(B * nh + ride_q->blocks, 1), dim3(NWQ * 64), 0, st, (const T*)qkv,
                                   (const T*)kr, xp, (const T*)psave, (const T*)dvec, (T*)gsave, (T*)dqkv, d_rwb, d_rrb, d_rsb, d_seg, L, nh, drop,
                                   *ride_q, B * nh);
                q_done = true;
            }
        }
        if (!q_done)
        hipLaunchKernelGGL((xl_attn_bwd_q_kernel<T, LP, NWQ>), dim3(B * nh, LP / 16 / NWQ), dim3(NWQ * 64), 0, st, (const T*)qkv, (const T*)kr,
                           xp, (const T*)psave, (const T*)dvec, (T*)gsave, (T*)dqkv, d_rwb, d_rrb, d_rsb, d_seg, L, nh, drop);
        if constexpr (sizeof(T) == 2 && LP <= 64) {
            if (g_kv2 && ride_kv != nullptr && ride_kv->blocks > 0 && ride_kv->n4 > 0) {
                gemm_log_ride(*ride_kv);
                hipLaunchKernelGGL((xl_attn_bwd_kv2_ride_kernel<T, LP, NWK>), dim3(B * nh + ride_kv->blocks), dim3(NWK * 64), 0, st, (const T*)qkv, xp,
                                   (const T*)psave, (const T*)gsave, (const T*)dvec, (T*)dqkv, (T*)dkr, L, nh, drop, *ride_kv, B * nh);
            } else if (g_kv2)
                hipLaunchKernelGGL((xl_attn_bwd_kv2_kernel<T, LP, NWK>), dim3(B * nh), dim3(NWK * 64), 0, st, (const T*)qkv, xp,
                                   (const T*)psave, (const T*)gsave, (const T*)dvec, (T*)dqkv, (T*)dkr, L, nh, drop);
            else
                hipLaunchKernelGGL((xl_attn_bwd_kv_kernel<T, LP, NWK>), dim3(B * nh), dim3(NWK * 64), 0, st, (const T*)qkv, xp,
                                   (const T*)psave, (const T*)gsave, (const T*)dvec, (T*)dqkv, (T*)dkr, L, nh, drop);
        } else
        hipLaunchKernelGGL((xl_attn_bwd_kv_kernel<T, LP, NWK>), dim3(B * nh), dim3(NWK * 64), 0, st, (const T*)qkv, xp,
                           (const T*)psave, (const T*)gsave, (const T*)dvec, (T*)dqkv, (T*)dkr, L, nh, drop);
    })
}

}  // namespace mb
