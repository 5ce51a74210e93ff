// Shared building blocks of the LDS-resident attention kernels (BERT: attention.hip, XLNet: xlnet_attention.hip).
#pragma once
#include "kernels.h"

namespace mb {

template <class T> struct AttnCfg;
template <> struct AttnCfg<bf16> { static constexpr int SLAB = 32, EPV = 8, ROWB = 128; };
template <> struct AttnCfg<float> { static constexpr int SLAB = 16, EPV = 4, ROWB = 256; };

template <class T>
__device__ __forceinline__ typename Frag<T>::type frag_nat(const char* img, int pitch, int row, int slab, int lane) {
    return *(const typename Frag<T>::type*)(img + row * pitch + slab * 64 + (lane >> 4) * 16);
}
// fragment of the TRANSPOSED image: element e = img[k0 + e][col].  Callers pass the MFMA operand pattern -- col = c0 + (lane & 15)
// with c0 a multiple of 16, k0 the same for the 16 lanes of a group, pitch and k0 * pitch multiples of 8 bytes -- which is what
// gfx950's transpose read serves: the 16 lanes of a group name a 4-row x 16-column block (lane t: row t >> 2, four columns from
// (t & 3) * 4) and each receives ITS column of the four rows.  Two ds_read_b64_tr_b16 per fragment instead of eight 2-byte
// reads and their packing.  (No LDS-DMA in the attention kernels, so the builtin's implicit vmcnt wait -- gemm.hip -- is harmless.)
__device__ __forceinline__ bf16x8 frag_kmaj(const char* img, int pitch, int k0, int col, bf16) {
    typedef __attribute__((ext_vector_type(4))) short s16x4_t;
    const int t = col & 15;
    const char* a = img + (k0 + (t >> 2)) * pitch + ((col - t) + (t & 3) * 4) * 2;
    union { s16x4_t h[2]; bf16x8 v; } u;
    u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(__attribute__((address_space(3))) char*)a);
    u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(__attribute__((address_space(3))) char*)(a + 4 * pitch));
    return u.v;
}
__device__ __forceinline__ f32x4 frag_kmaj(const char* img, int pitch, int k0, int col, float) {
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = *(const float*)(img + (k0 + e) * pitch + col * 4);
    return v;
}

// ---- products whose reduction index is the COLUMN index of an accumulator tile, without a round trip through LDS -----------
// After S^T-shaped tiles acc[t][r] = X[row = t*16 + (lane>>4)*4 + r][col = lane & 15] (mma16's layout) the next product sums over
// `row` (P.V over the keys, dS.K, Pd^T.dO, dS^T.Q).  An MFMA's k index is only a label: both operands just have to agree on which
// k slot holds which row.  So the lane's own accumulator values ARE its operand fragment -- bf16: slots 0..3 <- tile 2*sl, slots
// 4..7 <- tile 2*sl+1 (rows 2sl*16 + g*4 + r and (2sl+1)*16 + g*4 + r, g = lane >> 4); fp32 (16-wide slabs): tile sl as it is --
// and the k-major fragment of the other operand is read with the same row order (two transpose reads from two row groups).
// Before: every tile stored to a per-wave LDS strip, a barrier, and 16-byte reads back (3 strips and 6 barriers per backward strip).
template <class T> struct AccOp;
template <> struct AccOp<bf16> {
    static constexpr int TILES = 2;                                   // accumulator tiles per k-slab of 32
    static __device__ __forceinline__ bf16x8 make(const f32x4* t) {    // t[0], t[1]: tiles 2*sl, 2*sl+1
        bf16x8 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = from_f<bf16>(t[0][e]); v[4 + e] = from_f<bf16>(t[1][e]); }
        return v;
    }
    // element e of the fragment = img[row(e)][col], row(e) as above; col = c0 + (lane & 15), c0 a multiple of 16
    static __device__ __forceinline__ bf16x8 kmaj(const char* img, int pitch, int sl, int col, int lane) {
        typedef __attribute__((ext_vector_type(4))) short s16x4_t;
        const int t = col & 15, g = lane >> 4;
        const char* a = img + ((2 * sl) * 16 + g * 4 + (t >> 2)) * pitch + ((col - t) + (t & 3) * 4) * 2;
        union { s16x4_t h[2]; bf16x8 v; } u;
        u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(__attribute__((address_space(3))) char*)a);
        u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(__attribute__((address_space(3))) char*)(a + 16 * pitch));
        return u.v;
    }
};
template <> struct AccOp<float> {
    static constexpr int TILES = 1;
    static __device__ __forceinline__ f32x4 make(const f32x4* t) { return t[0]; }
    static __device__ __forceinline__ f32x4 kmaj(const char* img, int pitch, int sl, int col, int lane) {
        return frag_kmaj(img, pitch, sl * 16 + (lane >> 4) * 4, col, float());
    }
};

// stage rows [0, LP) x 64 elements of one head from the token-major tensor into an LDS image (rows >= L are zero)
template <class T, int LP, int NTHR>
__device__ __forceinline__ void stage_head(char* img, int pitch, const T* __restrict__ src, size_t ld, int L) {
    constexpr int CPR = AttnCfg<T>::ROWB / 16;
    for (int id = threadIdx.x; id < LP * CPR; id += NTHR) {
        const int row = id / CPR, c = id % CPR;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (row < L) v = *(const u32x4*)((const char*)(src + (size_t)row * ld) + c * 16);
        *(u32x4*)(img + row * pitch + c * 16) = v;
    }
}

// The same for N heads at once, every global load issued before the first LDS store: staged one after the other, each image
// was its own memory round trip (measured: 3.5 us of a 17 us attention backward before the first MFMA).
template <class T, int LP, int NTHR, int N>
__device__ __forceinline__ void stage_heads(char* const (&img)[N], int pitch, const T* const (&src)[N], const size_t (&ld)[N], int L) {
    constexpr int CPR = AttnCfg<T>::ROWB / 16;
    constexpr int IT = (LP * CPR + NTHR - 1) / NTHR;
    u32x4 v[N][IT];
#pragma unroll
    for (int n = 0; n < N; ++n)
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int id = threadIdx.x + it * NTHR, row = id / CPR, c = id % CPR;
            v[n][it] = u32x4{0u, 0u, 0u, 0u};
            if (id < LP * CPR && row < L) v[n][it] = *(const u32x4*)((const char*)(src[n] + (size_t)row * ld[n]) + c * 16);
        }
#pragma unroll
    for (int n = 0; n < N; ++n)
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int id = threadIdx.x + it * NTHR, row = id / CPR, c = id % CPR;
            if (id < LP * CPR) *(u32x4*)(img[n] + row * pitch + c * 16) = v[n][it];
        }
}

// N images with their own row counts (allocated rows RA[n], valid rows rows[n]); RMAX = max RA: MAG-XLNet stages the 2L-row
// relative-position keys next to the L-row q / k / v images.
template <class T, int NTHR, int N, int RMAX>
__device__ __forceinline__ void stage_heads_var(char* const (&img)[N], int pitch, const T* const (&src)[N], const size_t (&ld)[N],
                                                const int (&ralloc)[N], const int (&rows)[N]) {
    constexpr int CPR = AttnCfg<T>::ROWB / 16;
    constexpr int IT = (RMAX * CPR + NTHR - 1) / NTHR;
    u32x4 v[N][IT];
#pragma unroll
    for (int n = 0; n < N; ++n)
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int id = threadIdx.x + it * NTHR, row = id / CPR, c = id % CPR;
            v[n][it] = u32x4{0u, 0u, 0u, 0u};
            if (id < ralloc[n] * CPR && row < rows[n]) v[n][it] = *(const u32x4*)((const char*)(src[n] + (size_t)row * ld[n]) + c * 16);
        }
#pragma unroll
    for (int n = 0; n < N; ++n)
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int id = threadIdx.x + it * NTHR, row = id / CPR, c = id % CPR;
            if (id < ralloc[n] * CPR) *(u32x4*)(img[n] + row * pitch + c * 16) = v[n][it];
        }
}

// one image, plain loop (no register staging of the whole image: for the large images of the L = 128 XLNet kernels, where
// "every load first" would need hundreds of registers): rows [0, ralloc) x 64 elements, rows >= rows are zero
template <class T, int NTHR>
__device__ __forceinline__ void stage_rows(char* img, int pitch, const T* __restrict__ src, size_t ld, int ralloc, int rows) {
    constexpr int CPR = AttnCfg<T>::ROWB / 16;
    for (int id = threadIdx.x; id < ralloc * CPR; id += NTHR) {
        const int row = id / CPR, c = id % CPR;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (row < rows) v = *(const u32x4*)((const char*)(src + (size_t)row * ld) + c * 16);
        *(u32x4*)(img + row * pitch + c * 16) = v;
    }
}

// multiplies rows [0, LP) x 64 elements of a staged image by s (head_mask: every gradient of a head is linear in the head's dvec,
// so scaling the staged dvec image scales them all).  Call between two barriers.
template <class T, int LP, int NTHR>
__device__ __forceinline__ void scale_image(char* img, int pitch, float s) {
    for (int id = threadIdx.x; id < LP * 64; id += NTHR) {
        T* p = (T*)(img + (id >> 6) * pitch) + (id & 63);
        *p = from_f<T>(to_f(*p) * s);
    }
}

// row-wise reduction across the 4 lanes {i, i+16, i+32, i+48} that share a query/key row
__device__ __forceinline__ float quad_max(float v) {
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float quad_sum(float v) {
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}

constexpr float kMaskNeg = -10000.0f;     // transformers 3.0.2 get_extended_attention_mask
constexpr float kPadNeg = -1.0e30f;       // keys beyond L (tile padding only)


}  // namespace mb
