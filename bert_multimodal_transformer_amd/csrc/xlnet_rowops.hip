// Small HBM-bound kernels of the MAG-XLNet path (/root/reference/xlnet.py): word-embedding gather + dropout, the relative
// sinusoid table with its dropout, and the "last token" summary gather.
#include <algorithm>
#include "kernels.h"

namespace mb {

template <class T>
__global__ void __launch_bounds__(256) gather_drop_fwd_kernel(const int64_t* __restrict__ ids, const float* __restrict__ word,
                                                              T* __restrict__ out, int rows, int H, DropKey drop) {
    drop.resolve();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    if (row >= rows) return;
    const size_t id = ids ? (size_t)ids[row] : (size_t)row;      // ids == nullptr: inputs_embeds -- `word` is the [rows][H] embedding itself
    for (int col = lane * 4; col < H; col += 256) {
        f32x4 v = *(const f32x4*)(word + id * H + col);
        const uint32_t idx = (uint32_t)row * H + col;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] *= drop_mult(drop, idx + r);
        store4(out + (size_t)row * H + col, v);
    }
}

// Scatter-add of the embedding-output gradient into the word table.  XLNet inputs are LEFT padded with one id (<pad> = 5, which
// does receive a gradient: nn.Embedding without padding_idx, xlnet.py:28), so about half of all rows hit the same table row; one
// atomic per element serialises on it (measured 158 us).  Each thread owns one column of RC consecutive rows and merges runs of equal
// ids in a register before touching memory: the pad run of a sequence costs one atomic per column and chunk instead of one per row.
template <class T, int RC>
__global__ void __launch_bounds__(256) gather_drop_bwd_kernel(const T* __restrict__ dout, const int64_t* __restrict__ ids,
                                                              float* dword, int rows, int H, DropKey drop, GradAcc ga) {
    drop.resolve();
    const int col = blockIdx.y * 256 + threadIdx.x;
    if (col >= H) return;
    const int r0 = blockIdx.x * RC, r1 = min(rows, r0 + RC);
    if (ids == nullptr) {        // inputs_embeds: the gradient of row r is row r of the output (no table, nothing to merge)
        for (int r = r0; r < r1; ++r)
            dword[(size_t)r * H + col] = to_f(dout[(size_t)r * H + col]) * drop_mult(drop, (uint32_t)r * (uint32_t)H + (uint32_t)col);
        return;
    }
    size_t cur = (size_t)ids[r0];
    float acc = 0.f;
    for (int r = r0; r < r1; ++r) {
        const size_t id = (size_t)ids[r];
        const float v = to_f(dout[(size_t)r * H + col]) * drop_mult(drop, (uint32_t)r * (uint32_t)H + (uint32_t)col);
        if (id != cur) {
            grad_add(ga, dword + cur * H + col, acc);
            acc = 0.f;
            cur = id;
        }
        acc += v;
    }
    grad_add(ga, dword + cur * H + col, acc);
}

// pos_seq = arange(L, -L, -1): row p <-> position L - p ; freq d in [0, H/2): inv = 10000^(-2d/H) ; [sin | cos]
template <class T>
__global__ void __launch_bounds__(256) pos_emb_kernel(T* __restrict__ out, int B, int L, int H, DropKey drop) {
    drop.resolve();
    // one thread per (position, column): powf + sinf / cosf once, then the B copies, which differ only by their dropout masks (one
    // thread per ELEMENT evaluated the transcendentals B times: 30 us per step at B = 48)
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 2 * L * H) return;
    const int c = i % H, p = i / H;
    const int half = H / 2;
    const int d = c < half ? c : c - half;
    const float inv_freq = 1.0f / powf(10000.0f, (float)(2 * d) / (float)H);
    const float a = (float)(L - p) * inv_freq;
    const float v = c < half ? sinf(a) : cosf(a);
    for (int b = 0; b < B; ++b) {
        // reference index space: pos_emb is [2L, B, H] (xlnet.py:99-100,333) -> element ((p*B + b)*H + c)
        const float m = drop_mult(drop, (uint32_t)(((size_t)p * B + b) * H + c));
        out[((size_t)b * 2 * L + p) * H + c] = from_f<T>(v * m);
    }
}

template <class T>
__global__ void last_token_fwd_kernel(const T* __restrict__ x, T* __restrict__ xs, int B, int L, int H, DropKey drop) {
    drop.resolve();
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * H) return;
    const int b = i / H, c = i % H;
    const size_t src = ((size_t)b * L + (L - 1)) * H + c;
    xs[i] = from_f<T>(to_f(x[src]) * drop_mult(drop, (uint32_t)src));
}
template <class T>
__global__ void last_token_bwd_kernel(const T* __restrict__ dxs, T* __restrict__ dx, int B, int L, int H, DropKey drop) {
    drop.resolve();
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * H) return;
    const int b = i / H, c = i % H;
    const size_t dst = ((size_t)b * L + (L - 1)) * H + c;
    dx[dst] = from_f<T>(to_f(dxs[i]) * drop_mult(drop, (uint32_t)dst));
}

// y = x * dropout mask over a [rows][H] activation (element index = its offset): the final dropout of xlnet.py:396 on the whole
// sequence output (MAG_XLNetModel's return value) and on the gradient that comes back for it
template <class T>
__global__ void __launch_bounds__(256) drop_rows_kernel(const T* __restrict__ x, T* __restrict__ y, size_t n, DropKey drop) {
    drop.resolve();
    for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (size_t)gridDim.x * 1024) {
        f32x4 v = load4(x + i);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] *= drop_mult(drop, (uint32_t)(i + r));
        store4(y + i, v);
    }
}

#define MB_DISPATCH_T(dtype, ...)                                  \
    if ((dtype) == DT_BF16) { typedef bf16 T; __VA_ARGS__ }        \
    else if ((dtype) == DT_F32) { typedef float T; __VA_ARGS__ }   \
    else return MB_ERR_DTYPE;

int gather_drop_forward(int dtype, const int64_t* ids, const float* word, void* out, int rows, int H, DropKey drop, hipStream_t st) {
    if (rows <= 0) return MB_OK;
    if (H % 4) return MB_ERR_SHAPE;
    MB_DISPATCH_T(dtype, { hipLaunchKernelGGL((gather_drop_fwd_kernel<T>), dim3((rows + 3) / 4), dim3(256), 0, st, ids, word, (T*)out, rows, H, drop); })
    return (int)hipGetLastError();
}
int gather_drop_backward(int dtype, const void* dout, const int64_t* ids, float* dword, int rows, int H, DropKey drop, hipStream_t st, GradAcc acc) {
    if (rows <= 0) return MB_OK;
    constexpr int RC = 16;
    if (ids == nullptr) acc = GradAcc{nullptr, nullptr};          // inputs_embeds: plain stores into the [rows][H] gradient, no table
    MB_DISPATCH_T(dtype, { hipLaunchKernelGGL((gather_drop_bwd_kernel<T, RC>), dim3((rows + RC - 1) / RC, (H + 255) / 256), dim3(256), 0, st, (const T*)dout, ids, dword, rows, H, drop, acc); })
    return (int)hipGetLastError();
}
// dst[i] += src[i] (fp32, n % 4 == 0): the second k-half of the relative-position weight gradient joins the first (xlnet_engine.hip)
__global__ void __launch_bounds__(256) add_f32_kernel(float* __restrict__ dst, const float* __restrict__ src, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        f32x4 a = ((const f32x4*)dst)[i];
        a += ((const f32x4*)src)[i];
        ((f32x4*)dst)[i] = a;
    }
}
int add_f32(float* dst, const float* src, size_t n, hipStream_t st) {
    if (n == 0) return MB_OK;
    if ((n & 3) || (((uintptr_t)dst | (uintptr_t)src) & 15)) return MB_ERR_SHAPE;
    hipLaunchKernelGGL(add_f32_kernel, dim3((unsigned)std::min<size_t>((n / 4 + 255) / 256, 1024)), dim3(256), 0, st, dst, src, n / 4);
    return (int)hipGetLastError();
}
int drop_rows(int dtype, const void* x, void* y, int rows, int H, DropKey drop, hipStream_t st) {
    if (rows <= 0) return MB_OK;
    if (H % 4) return MB_ERR_SHAPE;
    const size_t n = (size_t)rows * H;
    MB_DISPATCH_T(dtype, { hipLaunchKernelGGL((drop_rows_kernel<T>), dim3((unsigned)std::min<size_t>((n / 4 + 255) / 256, 2048)), dim3(256), 0, st, (const T*)x, (T*)y, n, drop); })
    return (int)hipGetLastError();
}
int xlnet_pos_emb(int dtype, void* out, int B, int L, int H, DropKey drop, hipStream_t st) {
    MB_DISPATCH_T(dtype, { hipLaunchKernelGGL((pos_emb_kernel<T>), dim3((2 * L * H + 255) / 256), dim3(256), 0, st, (T*)out, B, L, H, drop); })
    return (int)hipGetLastError();
}
int last_token_forward(int dtype, const void* x, void* xs, int B, int L, int H, DropKey drop, hipStream_t st) {
    MB_DISPATCH_T(dtype, { hipLaunchKernelGGL((last_token_fwd_kernel<T>), dim3((B * H + 255) / 256), dim3(256), 0, st, (const T*)x, (T*)xs, B, L, H, drop); })
    return (int)hipGetLastError();
}
int last_token_backward(int dtype, const void* dxs, void* dx, int B, int L, int H, DropKey drop, hipStream_t st) {
    MB_DISPATCH_T(dtype, { hipLaunchKernelGGL((last_token_bwd_kernel<T>), dim3((B * H + 255) / 256), dim3(256), 0, st, (const T*)dxs, (T*)dx, B, L, H, drop); })
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------- query stream (xlnet.py:306-313, 374-399)
// g0[b][m][:] = mask_emb  (xlnet.py:306-310: word_emb_q = mask_emb.expand(M, B, -1); eval: the dropout behind it is the identity)
template <class T>
__global__ void __launch_bounds__(256) xl_broadcast_row_kernel(const float* __restrict__ row, T* __restrict__ out, int rows, int H) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = blockIdx.x * 4 + wave;
    if (r >= rows) return;
    for (int col = lane * 4; col < H; col += 256) store4(out + (size_t)r * H + col, *(const f32x4*)(row + col));
}
// The fused q | k | v operand of the query-stream attention.  Row (b, l): q = sum_m target_mapping[b][m][l] * qg[b][m][:]
// (einsum "mbnd,mlb->lbnd", modeling_xlnet's two-stream branch behind xlnet.py:374-385), k | v = the content stream's of the same
// layer.  fp32 accumulation over m, zero weights skipped (one-hot mappings read one row).
template <class T>
__global__ void __launch_bounds__(256) xl_map_q_kernel(const float* __restrict__ tm, const T* __restrict__ qg, const T* __restrict__ qkv_h,
                                                        T* __restrict__ qkv_g, int B, int M, int L, int H) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;              // b * L + l
    if (row >= B * L) return;
    const int b = row / L, l = row - b * L;
    const T* src = qkv_h + (size_t)row * 3 * H;
    T* dst = qkv_g + (size_t)row * 3 * H;
    for (int col = lane * 4; col < H; col += 256) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int m = 0; m < M; ++m) {
            const float w = tm[((size_t)b * M + m) * L + l];
            if (w != 0.f) acc += w * load4(qg + ((size_t)b * M + m) * H + col);
        }
        store4(dst + col, acc);
    }
    for (int col = H + lane * 4; col < 3 * H; col += 256) store4(dst + col, load4(src + col));
}
// vecg[b][m][:] = sum_l target_mapping[b][m][l] * vec[b][l][:]   (einsum "lbnd,mlb->mbnd")
template <class T>
__global__ void __launch_bounds__(256) xl_unmap_vec_kernel(const float* __restrict__ tm, const T* __restrict__ vec, T* __restrict__ vecg,
                                                            int B, int M, int L, int H) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;              // b * M + m
    if (row >= B * M) return;
    const int b = row / M;
    const float* w = tm + (size_t)row * L;
    for (int col = lane * 4; col < H; col += 256) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int l = 0; l < L; ++l) {
            const float wl = w[l];
            if (wl != 0.f) acc += wl * load4(vec + ((size_t)b * L + l) * H + col);
        }
        store4(vecg + (size_t)row * H + col, acc);
    }
}
int xlnet_broadcast_row(int dtype, const float* row, void* out, int rows, int H, hipStream_t st) {
    if (rows <= 0) return MB_OK;
    if (H % 4) return MB_ERR_SHAPE;
    MB_DISPATCH_T(dtype, { hipLaunchKernelGGL((xl_broadcast_row_kernel<T>), dim3((rows + 3) / 4), dim3(256), 0, st, row, (T*)out, rows, H); })
    return (int)hipGetLastError();
}
int xlnet_map_query(int dtype, const float* tm, const void* qg, const void* qkv_h, void* qkv_g, int B, int M, int L, int H, hipStream_t st) {
    if (H % 4) return MB_ERR_SHAPE;
    MB_DISPATCH_T(dtype, { hipLaunchKernelGGL((xl_map_q_kernel<T>), dim3((B * L + 3) / 4), dim3(256), 0, st, tm, (const T*)qg, (const T*)qkv_h, (T*)qkv_g, B, M, L, H); })
    return (int)hipGetLastError();
}
int xlnet_unmap_vec(int dtype, const float* tm, const void* vec, void* vecg, int B, int M, int L, int H, hipStream_t st) {
    if (H % 4) return MB_ERR_SHAPE;
    MB_DISPATCH_T(dtype, { hipLaunchKernelGGL((xl_unmap_vec_kernel<T>), dim3((B * M + 3) / 4), dim3(256), 0, st, tm, (const T*)vec, (T*)vecg, B, M, L, H); })
    return (int)hipGetLastError();
}

}  // namespace mb
