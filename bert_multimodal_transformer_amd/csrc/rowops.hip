// HBM-bound row kernels of the MAG-BERT path: LayerNorm fwd/bwd (with the dropout that follows or precedes
// it fused), BertEmbeddings gather+LN fwd / scatter bwd, column sums (bias grads), operand packing.
//
// Reference semantics:
//   BertEmbeddings / BertSelfOutput / BertOutput LayerNorm + dropout (transformers 3.0.2, called from
//   /root/reference/bert.py:211-229), MAG LayerNorm (/root/reference/modeling.py:22,47-49).
//
// Layout: one 64-lane wave owns one row of H = CH*256 elements; lane l holds CH chunks of 4 consecutive
// elements at columns (c*64 + l)*4  -> every global access is a coalesced 8/16-byte vector.
// Algorithmic HBM bytes per token are listed per kernel in DESIGN.md.
#include <algorithm>
#include "kernels.h"
#include "mag_pack.h"

namespace mb {

template <int CH>
struct RowStat {
    // two-pass mean / variance of a row held in registers (biased variance, like torch.nn.LayerNorm)
    static __device__ __forceinline__ void compute(const f32x4 (&v)[CH], float eps, float& mu, float& rs) {
        constexpr float invH = 1.0f / (CH * 256);
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c) s += (v[c][0] + v[c][1]) + (v[c][2] + v[c][3]);
        mu = wave_sum(s) * invH;
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float d = v[c][r] - mu; q += d * d; }
        }
        rs = 1.0f / sqrtf(wave_sum(q) * invH + eps);
    }
};

// ------------------------------------------------------------------------------------------ LayerNorm forward
template <class T, int CH>
__global__ void __launch_bounds__(256) ln_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float eps, T* __restrict__ y,
                                                     float* __restrict__ mean, float* __restrict__ rstd, int rows,
                                                     DropKey drop, Prefetch pf) {
    drop.resolve();
    constexpr int H = CH * 256;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    if (row >= rows) return;
    f32x4 v[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) v[c] = load4(x + (size_t)row * H + (c * 64 + lane) * 4);
    float mu, rs;
    RowStat<CH>::compute(v, eps, mu, rs);
    f32x4 g[CH], bt[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        g[c] = *(const f32x4*)(gamma + (c * 64 + lane) * 4);
        bt[c] = *(const f32x4*)(beta + (c * 64 + lane) * 4);
    }
    // behind the LAST of the kernel's own loads (loads return in order: anything issued after the prefetch would wait for it);
    // the stores below do not wait for loads, the wave does at its end
    u32x4 pfv[4];
    prefetch_issue<4>(pf, gamma, pfv);
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int col = (c * 64 + lane) * 4;
        f32x4 o = (v[c] - mu) * rs * g[c] + bt[c];
        const uint32_t idx = (uint32_t)row * H + col;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] *= drop_mult(drop, idx + r);
        store4(y + (size_t)row * H + col, o);
    }
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
    prefetch_retire<4>(pf, pfv);
}

#ifndef MB_LN_RPW
#define MB_LN_RPW 2
#endif
#ifndef MB_LN_NWV
#define MB_LN_NWV 8
#endif
constexpr int LN_RPW = MB_LN_RPW;     // rows per wave in the partial-sum variants (-DMB_LN_RPW / -DMB_LN_NWV: A/B builds)

// shared tail of the backward kernels: reduce per-lane column partials over the block's 4 waves, then atomics.
template <int CH, int NQ>
__device__ __forceinline__ void block_colsum_atomic(f32x4 (&part)[NQ][CH], float* const (&dst)[NQ], float* lds, GradAcc acc = {nullptr, nullptr}) {
    constexpr int H = CH * 256;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int c = 0; c < CH; ++c) *(f32x4*)(lds + (wave * NQ + q) * H + (c * 64 + lane) * 4) = part[q][c];
    __syncthreads();
    for (int i = threadIdx.x; i < NQ * H; i += 256) {
        const int q = i / H, col = i % H;
        if (dst[q] == nullptr) continue;
        const float s = lds[(0 * NQ + q) * H + col] + lds[(1 * NQ + q) * H + col] + lds[(2 * NQ + q) * H + col] +
                        lds[(3 * NQ + q) * H + col];
        grad_add(acc, dst[q] + col, s);
    }
}

// ------------------------------------------------------------------------------------------ LayerNorm backward
template <class T, int CH, int RPW, bool PARTIAL, int NWV = 4>
__global__ void __launch_bounds__(NWV * 64) ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, T* __restrict__ dx,
                                                     T* __restrict__ dx_drop, float* dgamma, float* dbeta, float* dbias,
                                                     int rows, DropKey drop_out, DropKey drop_in, Prefetch pf) {
    drop_out.resolve();
    drop_in.resolve();
    constexpr int H = CH * 256;
    constexpr float invH = 1.0f / H;
    constexpr int SL = NWV > 8 ? 8 : NWV;              // LDS slots of the block-level column sums (more waves fold in pairs first)
    __shared__ float lds[SL * 3 * H];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 part[3][CH];
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int c = 0; c < CH; ++c) part[q][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 g[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) g[c] = *(const f32x4*)(gamma + (c * 64 + lane) * 4);
    // (issuing the loads of all RPW rows ahead of the first reduction was measured: 12.8 vs 8.4 us per launch -- the rolled loop
    //  keeps the block at 124 VGPRs and lets the four waves drift apart, which overlaps their round trips better)
    for (int i = 0; i < RPW; ++i) {
        const int row = (blockIdx.x * NWV + wave) * RPW + i;
        if (row >= rows) break;
        const float mu = mean[row], rs = rstd[row];
        f32x4 dyv[CH], xh[CH];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int col = (c * 64 + lane) * 4;
            dyv[c] = load4(dy + (size_t)row * H + col);
            const uint32_t idx = (uint32_t)row * H + col;
#pragma unroll
            for (int r = 0; r < 4; ++r) dyv[c][r] *= drop_mult(drop_out, idx + r);
            xh[c] = (load4(x + (size_t)row * H + col) - mu) * rs;
            const f32x4 dxh = dyv[c] * g[c];
            s1 += (dxh[0] + dxh[1]) + (dxh[2] + dxh[3]);
            const f32x4 t = dxh * xh[c];
            s2 += (t[0] + t[1]) + (t[2] + t[3]);
        }
        const float m1 = wave_sum(s1) * invH, m2 = wave_sum(s2) * invH;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int col = (c * 64 + lane) * 4;
            f32x4 d = (dyv[c] * g[c] - m1 - xh[c] * m2) * rs;
            store4(dx + (size_t)row * H + col, d);
            part[0][c] += dyv[c] * xh[c];
            part[1][c] += dyv[c];
            if (dx_drop) {
                const uint32_t idx = (uint32_t)row * H + col;
#pragma unroll
                for (int r = 0; r < 4; ++r) d[r] *= drop_mult(drop_in, idx + r);
                store4(dx_drop + (size_t)row * H + col, d);
            }
            part[2][c] += d;
        }
    }
    u32x4 pfv[8];
    prefetch_issue<8>(pf, gamma, pfv);       // behind the last of the kernel's own loads; runs under the column-sum epilogue
    if constexpr (PARTIAL) {
        // dgamma points at partials[nblk][3][H]: this block's slab gets the sums over its waves, no atomics
        if constexpr (NWV > SL) {           // waves SL .. NWV-1 hand their sums to waves 0 .. SL-1 first
            if (wave >= SL) {
#pragma unroll
                for (int q = 0; q < 3; ++q)
#pragma unroll
                    for (int c = 0; c < CH; ++c) *(f32x4*)(lds + ((wave - SL) * 3 + q) * H + (c * 64 + lane) * 4) = part[q][c];
            }
            __syncthreads();
            if (wave < SL) {
#pragma unroll
                for (int q = 0; q < 3; ++q)
#pragma unroll
                    for (int c = 0; c < CH; ++c) part[q][c] += *(const f32x4*)(lds + (wave * 3 + q) * H + (c * 64 + lane) * 4);
            }
            __syncthreads();
        }
        if (wave < SL) {
#pragma unroll
            for (int q = 0; q < 3; ++q)
#pragma unroll
                for (int c = 0; c < CH; ++c) *(f32x4*)(lds + (wave * 3 + q) * H + (c * 64 + lane) * 4) = part[q][c];
        }
        __syncthreads();
        float* slab = dgamma + (size_t)blockIdx.x * 3 * H;
        for (int i = threadIdx.x; i < 3 * H; i += NWV * 64) {
            const int q = i / H, col = i % H;
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < SL; ++w) t += lds[(w * 3 + q) * H + col];
            slab[i] = t;
        }
    } else {
        float* const dst[3] = {dgamma, dbeta, dbias};
        static_assert(PARTIAL || NWV == 4, "block_colsum_atomic sums four waves");
        block_colsum_atomic<CH, 3>(part, dst, lds);
    }    prefetch_retire<8>(pf, pfv);
}

// out[q][col] += sum_b partials[b][q][col] for two partial sets (q = 0..2 from set a, 3..5 from set b).
// grid (H/64, 6, 8): the slabs are split 8 ways across blocks and 4 ways across the waves of a block, one atomic per
// column per block at the end (8-way contention).
__global__ void __launch_bounds__(256) ln_reduce_kernel(const float* __restrict__ pa, const float* __restrict__ pb, int nblk,
                                                        int H, float* d0, float* d1, float* d2, float* d3, float* d4,
                                                        float* d5, GradAcc acc) {
    const int q = blockIdx.y;
    const int col = blockIdx.x * 64 + (threadIdx.x & 63);
    const int part = (threadIdx.x >> 6) + 4 * blockIdx.z;    // 0 .. 4*gridDim.z-1
    const int nparts = 4 * gridDim.z;
    __shared__ float red[4][64];
    float* dsts[6] = {d0, d1, d2, d3, d4, d5};
    float* dst = dsts[q];
    const float* src = q < 3 ? pa : pb;
    float s = 0.f;
    if (dst != nullptr && src != nullptr && col < H) {
        const int qq = q % 3;
        for (int b = part; b < nblk; b += nparts) s += src[((size_t)b * 3 + qq) * H + col];
    }
    red[threadIdx.x >> 6][threadIdx.x & 63] = s;
    __syncthreads();
    if ((threadIdx.x >> 6) == 0 && dst != nullptr && src != nullptr && col < H)
        grad_add(acc, dst + col, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// all layers at once: blockIdx.z = layer * 8 + slab octant; a lane sums FOUR adjacent columns (16-byte loads: a wave reads 1 KB
// of a slab row; with one column per lane the launch was 24 us for 66 MB)
__global__ void __launch_bounds__(256) ln_reduce_layers_kernel(const float* __restrict__ pa, const float* __restrict__ pb,
                                                               size_t layer_stride, int nblk_all, int H, LnReduceDst dst, GradAcc acc) {
    const int q = blockIdx.y;
    const int layer = blockIdx.z >> 3;
    const int nblk = dst.nblk[layer] > 0 ? dst.nblk[layer] : nblk_all;      // slabs of this slot
    const int col = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
    const int part = (threadIdx.x >> 6) + 4 * (blockIdx.z & 7);
    constexpr int nparts = 32;
    __shared__ f32x4 red[4][64];
    float* out = dst.d[layer][q];
    const float* src = (q < 3 ? pa : pb) + (size_t)layer * layer_stride;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (out != nullptr && col < H) {
        const int qq = q % 3;
        for (int b = part; b < nblk; b += nparts) s += *(const f32x4*)(src + ((size_t)b * 3 + qq) * H + col);
    }
    red[threadIdx.x >> 6][threadIdx.x & 63] = s;
    __syncthreads();
    if ((threadIdx.x >> 6) == 0 && out != nullptr && col < H) {
        const f32x4 t = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];      // (ln_reduce_kernel's order)
#pragma unroll
        for (int r = 0; r < 4; ++r) grad_add(acc, out + col + r, t[r]);
    }
}

// ------------------------------------------------------------------------------------------ embeddings
template <class T, int CH>
__global__ void __launch_bounds__(256) embed_fwd_kernel(const int64_t* __restrict__ ids, const int64_t* __restrict__ seg,
                                                        const float* __restrict__ word, const float* __restrict__ pos,
                                                        const float* __restrict__ type, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps, T* __restrict__ out,
                                                        float* __restrict__ mean, float* __restrict__ rstd, int rows, int L,
                                                        DropKey drop, const int64_t* __restrict__ pos_ids) {
    drop.resolve();
    constexpr int H = CH * 256;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    if (row >= rows) return;
    // ids == nullptr: inputs_embeds (bert.py:185-195) -- `word` is then the [rows][H] fp32 embedding of each token itself
    // pos_ids == nullptr: BertEmbeddings' default position_ids = arange(L) (bert.py:211-216), i.e. the row's index in its sample
    const size_t id = ids ? (size_t)ids[row] : (size_t)row, sg = (size_t)seg[row], l = pos_ids ? (size_t)pos_ids[row] : (size_t)(row % L);
    f32x4 v[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int col = (c * 64 + lane) * 4;
        v[c] = *(const f32x4*)(word + id * H + col) + *(const f32x4*)(pos + l * H + col) +
               *(const f32x4*)(type + sg * H + col);
    }
    float mu, rs;
    RowStat<CH>::compute(v, eps, mu, rs);
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int col = (c * 64 + lane) * 4;
        const f32x4 g = *(const f32x4*)(gamma + col), b = *(const f32x4*)(beta + col);
        f32x4 o = (v[c] - mu) * rs * g + b;
        const uint32_t idx = (uint32_t)row * H + col;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] *= drop_mult(drop, idx + r);
        store4(out + (size_t)row * H + col, o);
    }
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
}

template <class T, int CH, int RPW, bool PARTIAL>
__global__ void __launch_bounds__(256) embed_bwd_kernel(const T* __restrict__ dout, const int64_t* __restrict__ ids,
                                                        const int64_t* __restrict__ seg, const float* __restrict__ word,
                                                        const float* __restrict__ pos, const float* __restrict__ type,
                                                        const float* __restrict__ gamma, const float* __restrict__ mean,
                                                        const float* __restrict__ rstd, float* __restrict__ dsum_ws,
                                                        float* dword, float* dgamma, float* dbeta, int rows, int L,
                                                        int pad_id, DropKey drop, const int64_t* __restrict__ pos_ids, GradAcc acc, int* id_count) {
    drop.resolve();
    constexpr int H = CH * 256;
    constexpr float invH = 1.0f / H;
    __shared__ float lds[4 * 2 * H];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 part[2][CH];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int c = 0; c < CH; ++c) part[q][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < RPW; ++i) {
        const int row = (blockIdx.x * 4 + wave) * RPW + i;
        if (row >= rows) break;
        const size_t id = ids ? (size_t)ids[row] : (size_t)row, sg = (size_t)seg[row], l = pos_ids ? (size_t)pos_ids[row] : (size_t)(row % L);
        const float mu = mean[row], rs = rstd[row];
        // (a stale 0 -- another row of the same id was faster and cleared the entry -- only selects the atomic path)
        const bool uniq = id_count != nullptr && ids != nullptr && acc.shadow == nullptr && id_count[id] == 1;
        f32x4 dyv[CH], xh[CH], gg[CH];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int col = (c * 64 + lane) * 4;
            const f32x4 xv = *(const f32x4*)(word + id * H + col) + *(const f32x4*)(pos + l * H + col) +
                             *(const f32x4*)(type + sg * H + col);
            xh[c] = (xv - mu) * rs;
            dyv[c] = load4(dout + (size_t)row * H + col);
            const uint32_t idx = (uint32_t)row * H + col;
#pragma unroll
            for (int r = 0; r < 4; ++r) dyv[c][r] *= drop_mult(drop, idx + r);
            gg[c] = *(const f32x4*)(gamma + col);
            const f32x4 dxh = dyv[c] * gg[c];
            s1 += (dxh[0] + dxh[1]) + (dxh[2] + dxh[3]);
            const f32x4 t = dxh * xh[c];
            s2 += (t[0] + t[1]) + (t[2] + t[3]);
        }
        const float m1 = wave_sum(s1) * invH, m2 = wave_sum(s2) * invH;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int col = (c * 64 + lane) * 4;
            const f32x4 d = (dyv[c] * gg[c] - m1 - xh[c] * m2) * rs;
            *(f32x4*)(dsum_ws + (size_t)row * H + col) = d;
            // dword == nullptr (inputs_embeds): the gradient of the given embeddings is dsum_ws itself
            if (dword != nullptr && (int)id != pad_id) {      // nn.Embedding(padding_idx=pad_token_id): no gradient for the pad row
                if (uniq) {                   // the only row of the batch with this id: nobody else touches its gradient row
                    f32x4* gp = (f32x4*)(dword + id * H + col);
                    *gp = *gp + d;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) grad_add(acc, dword + id * H + col + r, d[r]);
                }
            }
            part[0][c] += dyv[c] * xh[c];
            part[1][c] += dyv[c];
        }
        if (id_count != nullptr && ids != nullptr && lane == 0) id_count[id] = 0;      // (the table is all zero again after the backward)
    }
    if constexpr (PARTIAL) {
        // dgamma points at a partial set [nblk][3][H] (ln_bwd's layout; row 2 unused): this block's slab gets the 4-wave sums of
        // dgamma / dbeta, no atomics -- the step's ONE reduction launch (ln_reduce_partials_layers) adds them up
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int c = 0; c < CH; ++c) *(f32x4*)(lds + (wave * 2 + q) * H + (c * 64 + lane) * 4) = part[q][c];
        __syncthreads();
        float* slab = dgamma + (size_t)blockIdx.x * 3 * H;
        for (int i = threadIdx.x; i < 2 * H; i += 256) {
            const int q = i / H, col = i % H;
            slab[i] = lds[(0 * 2 + q) * H + col] + lds[(1 * 2 + q) * H + col] + lds[(2 * 2 + q) * H + col] + lds[(3 * 2 + q) * H + col];
        }
    } else {
        float* const dst[2] = {dgamma, dbeta};
        block_colsum_atomic<CH, 2>(part, dst, lds, acc);
    }
}

// The embedding backward of the engines' default case (position_ids = arange(L)): block (l, g) takes the rows of position l of
// samples [8g, 8g + 8) -- two per wave -- so that the position-table gradient of l is a sum over the block's own rows: no [T][H]
// fp32 round trip through dsum_ws and no second launch (embed_pos_type_kernel: 14 us) for dpos / dtype.  Five column sums per
// block: dgamma, dbeta, dtype[0] (set a of the slab, rows 0..2), dtype[1] (set b, row 0) -> the step's reduction launch; dpos[l]
// by one add per column per block (B / 8 blocks share a row).  dsum_ws is still written when it IS the result (inputs_embeds).
template <class T, int CH>
__global__ void __launch_bounds__(256) embed_bwd_fused_kernel(const T* __restrict__ dout, const int64_t* __restrict__ ids,
                                                              const int64_t* __restrict__ seg, const float* __restrict__ word,
                                                              const float* __restrict__ pos, const float* __restrict__ type,
                                                              const float* __restrict__ gamma, const float* __restrict__ mean,
                                                              const float* __restrict__ rstd, float* __restrict__ dsum_ws,
                                                              float* dword, float* dpos, float* dtype_, float* part_a, float* part_b,
                                                              int B, int L, int pad_id, DropKey drop, GradAcc acc, int* id_count) {
    drop.resolve();
    constexpr int H = CH * 256;
    constexpr float invH = 1.0f / H;
    __shared__ float lds[4 * 5 * H];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nbg = (B + 7) / 8;
    const int l = blockIdx.x / nbg, g = blockIdx.x % nbg;
    f32x4 part[5][CH];          // dgamma, dbeta, dtype0, dtype1, dpos
#pragma unroll
    for (int q = 0; q < 5; ++q)
#pragma unroll
        for (int c = 0; c < CH; ++c) part[q][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 2; ++i) {
        const int b = g * 8 + wave * 2 + i;
        if (b >= B) break;
        const int row = b * L + l;
        const size_t id = ids ? (size_t)ids[row] : (size_t)row, sg = (size_t)seg[row];
        const float mu = mean[row], rs = rstd[row];
        // (a stale 0 -- another row of the same id was faster and cleared the entry -- only selects the atomic path)
        const bool uniq = id_count != nullptr && ids != nullptr && acc.shadow == nullptr && id_count[id] == 1;
        f32x4 dyv[CH], xh[CH], gg[CH];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int col = (c * 64 + lane) * 4;
            const f32x4 xv = *(const f32x4*)(word + id * H + col) + *(const f32x4*)(pos + (size_t)l * H + col) +
                             *(const f32x4*)(type + sg * H + col);
            xh[c] = (xv - mu) * rs;
            dyv[c] = load4(dout + (size_t)row * H + col);
            const uint32_t idx = (uint32_t)row * H + col;
#pragma unroll
            for (int r = 0; r < 4; ++r) dyv[c][r] *= drop_mult(drop, idx + r);
            gg[c] = *(const f32x4*)(gamma + col);
            const f32x4 dxh = dyv[c] * gg[c];
            s1 += (dxh[0] + dxh[1]) + (dxh[2] + dxh[3]);
            const f32x4 t = dxh * xh[c];
            s2 += (t[0] + t[1]) + (t[2] + t[3]);
        }
        const float m1 = wave_sum(s1) * invH, m2 = wave_sum(s2) * invH;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int col = (c * 64 + lane) * 4;
            const f32x4 d = (dyv[c] * gg[c] - m1 - xh[c] * m2) * rs;
            if (ids == nullptr) *(f32x4*)(dsum_ws + (size_t)row * H + col) = d;      // inputs_embeds: this IS their gradient
            if (dword != nullptr && (int)id != pad_id) {      // nn.Embedding(padding_idx=pad_token_id): no gradient for the pad row
                if (uniq) {                   // the only row of the batch with this id: nobody else touches its gradient row
                    f32x4* gp = (f32x4*)(dword + id * H + col);
                    *gp = *gp + d;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) grad_add(acc, dword + id * H + col + r, d[r]);
                }
            }
            part[0][c] += dyv[c] * xh[c];
            part[1][c] += dyv[c];
            part[4][c] += d;
            if (sg == 0) part[2][c] += d;
            else if (sg == 1) part[3][c] += d;
            else {
#pragma unroll
                for (int r = 0; r < 4; ++r) grad_add(acc, dtype_ + sg * H + col + r, d[r]);     // (type_vocab_size > 2: rare rows)
            }
        }
        if (id_count != nullptr && ids != nullptr && lane == 0) id_count[id] = 0;      // (the table is all zero again after the backward)
    }
#pragma unroll
    for (int q = 0; q < 5; ++q)
#pragma unroll
        for (int c = 0; c < CH; ++c) *(f32x4*)(lds + (wave * 5 + q) * H + (c * 64 + lane) * 4) = part[q][c];
    __syncthreads();
    float* slab_a = part_a + (size_t)blockIdx.x * 3 * H;
    float* slab_b = part_b + (size_t)blockIdx.x * 3 * H;
    for (int i = threadIdx.x; i < 5 * H; i += 256) {
        const int q = i / H, col = i % H;
        const float t = lds[(0 * 5 + q) * H + col] + lds[(1 * 5 + q) * H + col] + lds[(2 * 5 + q) * H + col] + lds[(3 * 5 + q) * H + col];
        if (q < 3) slab_a[q * H + col] = t;
        else if (q == 3) slab_b[col] = t;
        else grad_add(acc, dpos + (size_t)l * H + col, t);
    }
}

// position / token-type table grads: block (l, c) sums 256 columns of dsum over the batch (sole owner of that piece of dpos[l]).
__global__ void __launch_bounds__(256) embed_pos_type_kernel(const float* __restrict__ dsum_ws, const int64_t* __restrict__ seg,
                                                             float* dpos, float* dtype_, int B, int L, int H,
                                                             const int64_t* __restrict__ pos_ids, GradAcc acc) {
    const int l = blockIdx.x;
    {
        const int col = blockIdx.y * 256 + threadIdx.x;
        if (col >= H) return;
        float ap = 0.f, a0 = 0.f, a1 = 0.f;
        // eight samples per batch of loads (one sample per iteration made the sweep over B a chain of B memory round trips)
        for (int b0 = 0; b0 < B; b0 += 8) {
            float d[8];
            int64_t sg[8], pp[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int b = b0 + u;
                const bool ok = b < B;
                const int t = (ok ? b : 0) * L + l;
                d[u] = ok ? dsum_ws[(size_t)t * H + col] : 0.f;
                sg[u] = ok ? seg[t] : 0;
                pp[u] = (ok && pos_ids) ? pos_ids[t] : -1;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (pp[u] >= 0) grad_add(acc, dpos + (size_t)pp[u] * H + col, d[u]);      // explicit position_ids: rows of other blocks too
                else ap += d[u];
                if (sg[u] == 0) a0 += d[u];
                else if (sg[u] == 1) a1 += d[u];
                else grad_add(acc, dtype_ + (size_t)sg[u] * H + col, d[u]);
            }
        }
        if (!pos_ids) dpos[(size_t)l * H + col] += ap;
        grad_add(acc, dtype_ + col, a0);
        grad_add(acc, dtype_ + H + col, a1);
    }
}

// ------------------------------------------------------------------------------------------ column sum
template <class T>
__global__ void __launch_bounds__(256) colsum_kernel(const T* __restrict__ x, int ldx, float* out, int rows, int cols,
                                                     int rows_per_block, GradAcc gacc) {
    __shared__ f32x4 lds[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = (blockIdx.x * 64 + lane) * 4;
    const int r0 = blockIdx.y * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (col < cols)
        for (int r = r0 + wave; r < r1; r += 4) acc += load4(x + (size_t)r * ldx + col);
    lds[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && col < cols) {
        const f32x4 s = lds[0][lane] + lds[1][lane] + lds[2][lane] + lds[3][lane];
#pragma unroll
        for (int r = 0; r < 4; ++r) grad_add(gacc, out + col + r, s[r]);
    }
}

template <class T>
__global__ void pack_pad_kernel(const float* __restrict__ src, int cols, T* __restrict__ dst, int cols_pad, size_t total) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const size_t r = i / cols_pad;
    const int c = (int)(i % cols_pad);
    dst[i] = from_f<T>(c < cols ? src[r * cols + c] : 0.f);
}

template <class T>
__global__ void convert_kernel(const float* __restrict__ src, T* __restrict__ dst, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256)
        store4(dst + i * 4, *(const f32x4*)(src + i * 4));
}

template <class T>
__global__ void widen_kernel(const T* __restrict__ src, float* __restrict__ dst, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256)
        *(f32x4*)(dst + i * 4) = load4(src + i * 4);
}

// ------------------------------------------------------------------------------------------ host launchers
#define MB_DISPATCH_T(dtype, ...)                                  \
    if ((dtype) == DT_BF16) { typedef bf16 T; __VA_ARGS__ }        \
    else if ((dtype) == DT_F32) { typedef float T; __VA_ARGS__ }   \
    else return MB_ERR_DTYPE;
#define MB_DISPATCH_CH(H, ...)                                     \
    if ((H) == 768) { constexpr int CH = 3; __VA_ARGS__ }          \
    else if ((H) == 1024) { constexpr int CH = 4; __VA_ARGS__ }    \
    else if ((H) == 512) { constexpr int CH = 2; __VA_ARGS__ }     \
    else if ((H) == 256) { constexpr int CH = 1; __VA_ARGS__ }     \
    else return MB_ERR_SHAPE;

int ln_forward(int dtype, const void* x, const float* gamma, const float* beta, float eps, void* y, float* mean,
               float* rstd, int rows, int H, DropKey drop, hipStream_t st, Prefetch pf) {
    if (rows <= 0) return MB_OK;
    MB_DISPATCH_T(dtype, MB_DISPATCH_CH(H, {
        hipLaunchKernelGGL((ln_fwd_kernel<T, CH>), dim3((rows + 3) / 4), dim3(256), 0, st, (const T*)x, gamma, beta, eps,
                           (T*)y, mean, rstd, rows, drop, pf);
    }))
    return (int)hipGetLastError();
}

int ln_backward(int dtype, const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                void* dx, void* dx_drop, float* dgamma, float* dbeta, float* dbias, int rows, int H, DropKey drop_out,
                DropKey drop_in, hipStream_t st) {
    if (rows <= 0) return MB_OK;
    constexpr int RPW = 4;
    MB_DISPATCH_T(dtype, MB_DISPATCH_CH(H, {
        hipLaunchKernelGGL((ln_bwd_kernel<T, CH, RPW, false>), dim3((rows + 4 * RPW - 1) / (4 * RPW)), dim3(256), 0, st,
                           (const T*)dy, (const T*)x, gamma, mean, rstd, (T*)dx, (T*)dx_drop, dgamma, dbeta, dbias, rows,
                           drop_out, drop_in, Prefetch{nullptr, 0, nullptr});
    }))
    return (int)hipGetLastError();
}

size_t ln_partials_floats(int rows, int H) { return (size_t)((rows + 4 * LN_RPW - 1) / (4 * LN_RPW)) * 3 * H; }

// rows per block of the partial-sum LayerNorm backward: 8 waves x LN_RPW rows (half the slabs of a 4-wave block -- the reduction
// launch reads them all -- at the same number of waves in flight)
constexpr int LN_NWV = MB_LN_NWV;
int ln_backward_partials(int dtype, const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                         void* dx, void* dx_drop, float* partials, int* nblk, int rows, int H, DropKey drop_in,
                         hipStream_t st, Prefetch pf) {
    if (rows <= 0) { *nblk = 0; return MB_OK; }
    const int nb = (rows + LN_NWV * LN_RPW - 1) / (LN_NWV * LN_RPW);
    *nblk = nb;
    const DropKey nodrop = {0u, 0u, 0u, 1.0f};
    MB_DISPATCH_T(dtype, MB_DISPATCH_CH(H, {
        hipLaunchKernelGGL((ln_bwd_kernel<T, CH, LN_RPW, true, LN_NWV>), dim3(nb), dim3(LN_NWV * 64), 0, st, (const T*)dy, (const T*)x, gamma,
                           mean, rstd, (T*)dx, (T*)dx_drop, partials, (float*)nullptr, (float*)nullptr, rows, nodrop, drop_in, pf);
    }))
    return (int)hipGetLastError();
}

int ln_reduce_partials(const float* pa, const float* pb, int nblk, int H, float* const* d, hipStream_t st, GradAcc acc) {
    if (nblk <= 0) return MB_OK;
    hipLaunchKernelGGL(ln_reduce_kernel, dim3((H + 63) / 64, 6, 8), dim3(256), 0, st, pa, pb, nblk, H, d[0], d[1], d[2], d[3], d[4],
                       d[5], acc);
    return (int)hipGetLastError();
}

int ln_reduce_partials_layers(const float* pa, const float* pb, size_t layer_stride, int layers, int nblk, int H,
                              const LnReduceDst& dst, hipStream_t st, GradAcc acc) {
    if (nblk <= 0 || layers <= 0) return MB_OK;
    if (layers > MB_LN_MAX_LAYERS) return MB_ERR_SHAPE;
    if (H % 4) return MB_ERR_SHAPE;
    hipLaunchKernelGGL(ln_reduce_layers_kernel, dim3((H + 255) / 256, 6, 8 * layers), dim3(256), 0, st, pa, pb, layer_stride, nblk, H, dst, acc);
    return (int)hipGetLastError();
}

int embed_ln_forward(int dtype, const int64_t* ids, const int64_t* seg, const float* word, const float* pos,
                     const float* type, const float* gamma, const float* beta, float eps, void* out, float* mean,
                     float* rstd, int B, int L, int H, DropKey drop, hipStream_t st, const int64_t* pos_ids) {
    const int rows = B * L;
    if (rows <= 0) return MB_OK;
    MB_DISPATCH_T(dtype, MB_DISPATCH_CH(H, {
        hipLaunchKernelGGL((embed_fwd_kernel<T, CH>), dim3((rows + 3) / 4), dim3(256), 0, st, ids, seg, word, pos, type,
                           gamma, beta, eps, (T*)out, mean, rstd, rows, L, drop, pos_ids);
    }))
    return (int)hipGetLastError();
}

int embed_ln_backward(int dtype, const void* dout, const int64_t* ids, const int64_t* seg, const float* word,
                      const float* pos, const float* type, const float* gamma, const float* mean, const float* rstd,
                      float* dsum_ws, float* dword, float* dpos, float* dtype_, float* dgamma, float* dbeta, int B, int L,
                      int H, int pad_id, DropKey drop, hipStream_t st, const int64_t* pos_ids, GradAcc acc, float* part, int* nblk,
                      float* part_b, int* id_count) {
    const int rows = B * L;
    if (nblk) *nblk = (rows + 4 * LN_RPW - 1) / (4 * LN_RPW);
    if (rows <= 0) return MB_OK;
    if (part != nullptr && part_b != nullptr && pos_ids == nullptr) {
        // default positions: one launch for everything (embed_bwd_fused_kernel); L * ceil(B / 8) slabs in both sets
        if (nblk) *nblk = L * ((B + 7) / 8);
        MB_DISPATCH_T(dtype, MB_DISPATCH_CH(H, {
            hipLaunchKernelGGL((embed_bwd_fused_kernel<T, CH>), dim3(L * ((B + 7) / 8)), dim3(256), 0, st, (const T*)dout, ids, seg, word,
                               pos, type, gamma, mean, rstd, dsum_ws, dword, dpos, dtype_, part, part_b, B, L, pad_id, drop, acc, id_count);
        }))
        return (int)hipGetLastError();
    }
    if (part != nullptr) {       // dgamma / dbeta as one slab per block of 4 * LN_RPW rows (explicit position_ids: dpos / dtype by the second launch)
        MB_DISPATCH_T(dtype, MB_DISPATCH_CH(H, {
            hipLaunchKernelGGL((embed_bwd_kernel<T, CH, LN_RPW, true>), dim3((rows + 4 * LN_RPW - 1) / (4 * LN_RPW)), dim3(256), 0, st,
                               (const T*)dout, ids, seg, word, pos, type, gamma, mean, rstd, dsum_ws, dword, part, (float*)nullptr,
                               rows, L, pad_id, drop, pos_ids, acc, id_count);
        }))
    } else {
    constexpr int RPW = 4;
    MB_DISPATCH_T(dtype, MB_DISPATCH_CH(H, {
        hipLaunchKernelGGL((embed_bwd_kernel<T, CH, RPW, false>), dim3((rows + 4 * RPW - 1) / (4 * RPW)), dim3(256), 0, st,
                           (const T*)dout, ids, seg, word, pos, type, gamma, mean, rstd, dsum_ws, dword, dgamma, dbeta,
                           rows, L, pad_id, drop, pos_ids, acc, id_count);
    }))
    }
    hipLaunchKernelGGL(embed_pos_type_kernel, dim3(L, (H + 255) / 256), dim3(256), 0, st, dsum_ws, seg, dpos, dtype_, B, L, H, pos_ids, acc);
    return (int)hipGetLastError();
}

int colsum(int dtype, const void* x, int ldx, float* out, int rows, int cols, hipStream_t st, GradAcc acc) {
    if (rows <= 0 || cols <= 0) return MB_OK;
    if (cols % 4) return MB_ERR_SHAPE;
    const int cblocks = (cols + 255) / 256;
    int rblocks = (512 + cblocks - 1) / cblocks;
    if (rblocks > (rows + 15) / 16) rblocks = (rows + 15) / 16;
    if (rblocks < 1) rblocks = 1;
    const int rpb = (rows + rblocks - 1) / rblocks;
    rblocks = (rows + rpb - 1) / rpb;
    MB_DISPATCH_T(dtype, {
        hipLaunchKernelGGL((colsum_kernel<T>), dim3(cblocks, rblocks), dim3(256), 0, st, (const T*)x, ldx, out, rows,
                           cols, rpb, acc);
    })
    return (int)hipGetLastError();
}

int pack_pad(int dtype, const float* src, int cols, void* dst, int cols_pad, int rows, hipStream_t st) {
    const size_t total = (size_t)rows * cols_pad;
    if (total == 0) return MB_OK;
    MB_DISPATCH_T(dtype, {
        hipLaunchKernelGGL((pack_pad_kernel<T>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, src, cols,
                           (T*)dst, cols_pad, total);
    })
    return (int)hipGetLastError();
}

int convert(int dtype, const float* src, void* dst, size_t n, hipStream_t st) {
    if (n == 0) return MB_OK;
    if (n % 4) return MB_ERR_SHAPE;
    const size_t n4 = n / 4;
    unsigned grid = (unsigned)((n4 + 255) / 256);
    if (grid > 4096) grid = 4096;
    MB_DISPATCH_T(dtype, { hipLaunchKernelGGL((convert_kernel<T>), dim3(grid), dim3(256), 0, st, src, (T*)dst, n4); })
    return (int)hipGetLastError();
}

int widen(int dtype, const void* src, float* dst, size_t n, hipStream_t st) {
    if (n == 0) return MB_OK;
    if (n % 4) return MB_ERR_SHAPE;
    const size_t n4 = n / 4;
    unsigned grid = (unsigned)((n4 + 255) / 256);
    if (grid > 4096) grid = 4096;
    MB_DISPATCH_T(dtype, { hipLaunchKernelGGL((widen_kernel<T>), dim3(grid), dim3(256), 0, st, (const T*)src, dst, n4); })
    return (int)hipGetLastError();
}


// ------------------------------------------------------------------------------------------ zero fill
// The passes clear their scratch (packed MAG weight gradients, pad rows, the [CLS]-only pooler gradient slab, the loss word)
// with this launch instead of hipMemsetAsync: a captured step then consists of kernel nodes only (memset nodes of a replayed
// graph were the one thing that did not reproduce the eager result on ROCm 7.2: stale packed MAG gradients on replays).
__global__ void __launch_bounds__(256) zero_fill_kernel(const ZeroRanges z) {
    const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, nth = (size_t)gridDim.x * 256;
#pragma unroll 1
    for (int r = 0; r < z.n; ++r) {
        uint32_t* __restrict__ p = z.p[r];
        const size_t ndw = z.ndw[r];
        if (((uintptr_t)p & 15) == 0) {
            const size_t n4 = ndw / 4;
            const u32x4 zero = {0u, 0u, 0u, 0u};
            for (size_t i = tid; i < n4; i += nth) ((u32x4*)p)[i] = zero;
            for (size_t i = n4 * 4 + tid; i < ndw; i += nth) p[i] = 0u;
        } else {
            for (size_t i = tid; i < ndw; i += nth) p[i] = 0u;
        }
    }
}
int zero_fill_ranges(const ZeroRanges& z, hipStream_t st) {
    if (z.n < 0 || z.n > MB_ZERO_MAX) return MB_ERR_ARG;
    size_t most = 0;
    ZeroRanges c = {};
    for (int r = 0; r < z.n; ++r) {
        if (z.ndw[r] == 0) continue;
        if (!z.p[r] || ((uintptr_t)z.p[r] & 3)) return MB_ERR_ARG;
        c.p[c.n] = z.p[r]; c.ndw[c.n] = z.ndw[r]; ++c.n;
        if (z.ndw[r] > most) most = z.ndw[r];
    }
    if (c.n == 0) return MB_OK;
    unsigned grid = (unsigned)((most / 4 + 255) / 256);
    if (grid < 1) grid = 1;
    if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(zero_fill_kernel, dim3(grid), dim3(256), 0, st, c);
    return (int)hipGetLastError();
}
int zero_fill(void* p, size_t bytes, hipStream_t st) {
    if (bytes == 0) return MB_OK;
    if (!p || (bytes & 3)) return MB_ERR_ARG;
    ZeroRanges z = {};
    z.n = 1; z.p[0] = (uint32_t*)p; z.ndw[0] = bytes / 4;
    return zero_fill_ranges(z, st);
}

// ------------------------------------------------------------------------------------------ step prologue
// Device twin of make_key() (engine_common.h): k = splitmix64(splitmix64(seed) ^ splitmix64(step * FNV + site)).
__device__ __forceinline__ uint64_t splitmix64_dev(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__global__ void __launch_bounds__(256) step_prologue_kernel(const PrologueArgs a) {
    if ((int)blockIdx.x >= a.copy_blocks) {         // the extra blocks: MAG's weight operands (device memory only)
        const size_t first = (size_t)(blockIdx.x - a.copy_blocks) * 256 + threadIdx.x, stride = (size_t)(gridDim.x - a.copy_blocks) * 256;
        const PrologueArgs::MagPackW& w = a.magw;
        if (w.dtype == DT_BF16) mag_pack_w_range<bf16>(w.W_hv, w.W_ha, w.W_v, w.W_a, (bf16*)w.We, (bf16*)w.Wv, (bf16*)w.Wa, w.d, first, stride);
        else mag_pack_w_range<float>(w.W_hv, w.W_ha, w.W_v, w.W_a, (float*)w.We, (float*)w.Wv, (float*)w.Wa, w.d, first, stride);
        return;
    }
    const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, nth = (size_t)a.copy_blocks * 256;
    if (blockIdx.x == 0) {
        for (int s = threadIdx.x; s < a.nsites; s += 256) {
            const uint64_t h = splitmix64_dev(splitmix64_dev(a.seed) ^ splitmix64_dev(a.step * 0x100000001B3ull + (uint64_t)s));
            a.keys[2 * s] = (uint32_t)h;
            a.keys[2 * s + 1] = (uint32_t)(h >> 32);
        }
        if (a.adam_dst && threadIdx.x < 2) a.adam_dst[threadIdx.x] = a.adam[threadIdx.x];
    }
    if (tid == 0 && a.zero_dw) *a.zero_dw = 0u;
    // occurrences of every token id of this batch (a vocab-sized table that is all zero between steps): the embedding backward adds
    // the gradient row of an id that occurs ONCE with plain 16-byte read-modify-writes instead of four atomics per lane.  Counted
    // here because this launch waits for PCIe anyway (inside embed_fwd the 2,400 device-scope atomics cost 10 us)
    if (a.id_count != nullptr)
        for (size_t i = tid; i < (size_t)a.n_ids; i += nth) atomicAdd(a.id_count + a.ids[i], 1);
    for (int k = 0; k < a.npack; ++k) {          // (before the copies: these loads cross PCIe too and should be in flight with them)
        const PrologueArgs::PackJob& j = a.pack[k];
        const size_t n = (size_t)j.rows * j.cols;
        if ((((uintptr_t)j.src) & 15) == 0) {
            for (size_t i = tid; i < (n + 3) / 4; i += nth) {
                const size_t e0 = i * 4;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (e0 + 4 <= n) v = __builtin_nontemporal_load((const f32x4*)j.src + i);
                else for (int r = 0; r < 4; ++r) if (e0 + r < n) v[r] = j.src[e0 + r];
                size_t row = e0 / j.cols;
                int c = (int)(e0 - row * j.cols);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (e0 + r < n) {
                        const size_t d = row * j.pitch + c;
                        if (j.dtype == DT_BF16) ((bf16*)j.dst)[d] = from_f<bf16>(v[r]); else ((float*)j.dst)[d] = v[r];
                    }
                    if (++c == j.cols) { c = 0; ++row; }
                }
            }
        } else {
            for (size_t e = tid; e < n; e += nth) {
                const size_t row = e / j.cols, d = row * j.pitch + (e - row * j.cols);
                if (j.dtype == DT_BF16) ((bf16*)j.dst)[d] = from_f<bf16>(j.src[e]); else ((float*)j.dst)[d] = j.src[e];
            }
        }
    }
    // The sources may be pinned host memory: every 16-byte load of a pass is issued before the first store, so a thread pays ONE
    // round trip across PCIe for its pieces of all copies (copy after copy: one round trip per copy, 6 per step).
    size_t most4 = 0;
    bool fast = true;
    for (int c = 0; c < MB_PROLOGUE_MAX_COPIES; ++c)
        if (c < a.ncopies) {
            fast = fast && ((((uintptr_t)a.src[c] | (uintptr_t)a.dst[c]) & 15) == 0);
            if (a.dwords[c] / 4 > most4) most4 = a.dwords[c] / 4;
        }
    if (fast) {                                     // uniform
        for (size_t i = tid; i < most4; i += nth) {
            u32x4 v[MB_PROLOGUE_MAX_COPIES];
#pragma unroll
            for (int c = 0; c < MB_PROLOGUE_MAX_COPIES; ++c)
                if (c < a.ncopies && i < a.dwords[c] / 4) v[c] = __builtin_nontemporal_load((const u32x4*)a.src[c] + i);
#pragma unroll
            for (int c = 0; c < MB_PROLOGUE_MAX_COPIES; ++c)
                if (c < a.ncopies && i < a.dwords[c] / 4) ((u32x4*)a.dst[c])[i] = v[c];
        }
#pragma unroll 1
        for (int c = 0; c < a.ncopies; ++c) {
            const size_t n = a.dwords[c];
            for (size_t i = (n / 4) * 4 + tid; i < n; i += nth) a.dst[c][i] = a.src[c][i];
        }
        return;
    }
#pragma unroll 1
    for (int c = 0; c < a.ncopies; ++c) {
        const uint32_t* __restrict__ src = a.src[c];
        uint32_t* __restrict__ dst = a.dst[c];
        const size_t n = a.dwords[c];
        for (size_t i = tid; i < n; i += nth) dst[i] = src[i];
    }
}

int step_prologue(const PrologueArgs& a, hipStream_t st) {
    if (a.ncopies < 0 || a.ncopies > MB_PROLOGUE_MAX_COPIES || a.nsites < 0 || (a.nsites > 0 && !a.keys)) return MB_ERR_ARG;
    if (a.npack < 0 || a.npack > 2) return MB_ERR_ARG;
    size_t most = 0;
    for (int k = 0; k < a.npack; ++k) {
        if (!a.pack[k].src || !a.pack[k].dst || a.pack[k].cols < 1 || a.pack[k].pitch < a.pack[k].cols) return MB_ERR_ARG;
        most = std::max(most, (size_t)a.pack[k].rows * a.pack[k].cols);
    }
    for (int c = 0; c < a.ncopies; ++c) {
        if (a.dwords[c] && (!a.src[c] || !a.dst[c])) return MB_ERR_ARG;
        if (a.dwords[c] > most) most = a.dwords[c];
    }
    unsigned grid = (unsigned)((most / 4 + 255) / 256);
    if (grid < 1) grid = 1;
    if (grid > 512) grid = 512;
    PrologueArgs b = a;
    b.copy_blocks = (int)grid;
    if (a.magw.W_hv != nullptr) {
        if (!a.magw.W_ha || !a.magw.W_v || !a.magw.W_a || !a.magw.We || !a.magw.Wv || !a.magw.Wa) return MB_ERR_ARG;
        if (a.magw.dtype != DT_BF16 && a.magw.dtype != DT_F32) return MB_ERR_DTYPE;
        grid += 256;                                // one more block per CU, next to the ones that sit on their PCIe round trips
    }
    hipLaunchKernelGGL(step_prologue_kernel, dim3(grid), dim3(256), 0, st, b);
    return (int)hipGetLastError();
}


// ------------------------------------------------------------------------------------------ deterministic mode: shadow -> gradients
__global__ void __launch_bounds__(256) grad_fold_kernel(long long* __restrict__ shadow, float* __restrict__ g, size_t begin, size_t end) {
    for (size_t i = begin + (size_t)blockIdx.x * 256 + threadIdx.x; i < end; i += (size_t)gridDim.x * 256) {
        const long long v = shadow[i];
        if (v != 0) {
            g[i] += (float)((double)v * (1.0 / (double)kGradFix));
            shadow[i] = 0;
        }
    }
}
int grad_fold(GradAcc acc, float* g, size_t begin, size_t end, hipStream_t st) {
    if (!acc.shadow || end <= begin) return MB_OK;
    if (g != acc.base) return MB_ERR_ARG;
    const size_t n = end - begin;
    hipLaunchKernelGGL(grad_fold_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 4096)), dim3(256), 0, st, acc.shadow, g, begin, end);
    return (int)hipGetLastError();
}

}  // namespace mb
