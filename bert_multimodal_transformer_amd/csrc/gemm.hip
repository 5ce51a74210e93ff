// MFMA GEMM for the MAG-BERT encoder: C[M,N] = sum_k A(m,k) * B(n,k), with fused epilogues.
//
// Replaces the cuBLAS addmm / mm calls that torch.nn.Linear issues under the reference's
// BertSelfAttention/BertSelfOutput/BertIntermediate/BertOutput (transformers 3.0.2, called from
// /root/reference/bert.py:221-229) and MAG's four Linears (/root/reference/modeling.py:15-19,27-30),
// forward, dgrad and wgrad.
//
// Operand layouts (per operand, chosen at compile time):
//   row  : stored [rows][K], K contiguous   (X and W in  Y = X W^T ; dY in  dX = dY W)
//   kmaj : stored [K][rows], rows contiguous (W in dX = dY W ; dY and X in dW = dY^T X)
// Both end up in LDS as [row][k] images with k contiguous (144-byte pitch: 128 B of k + 16 B pad),
// so every MFMA fragment is one ds_read_b128.  kmaj operands are transposed in registers while staging
// (8 (bf16) / 4 (fp32) coalesced dword loads per lane -> one 16-byte LDS write per output row).
//
// Tile: BM x BN x 128 bytes-of-k (64 bf16 / 32 fp32), 256 threads = 2x2 waves, each wave (BM/2)x(BN/2)
// as MFMA 16x16 tiles; LDS double buffered, global loads for tile t+1 are in flight while tile t is
// multiplied (one barrier per k-tile).  fp32 uses v_mfma_f32_16x16x4_f32 (exact fp32 fma chain): this is
// the "fp32 parity mode" (north_star: logits within 1e-3 of the CPU reference); bf16 is the perf mode.
//
// The accumulator is computed transposed (mma16(acc, Bfrag, Afrag)) so that a lane owns 4 CONSECUTIVE
// columns n of one row m: epilogue loads/stores are 8/16-byte vectors and bias is one float4.
#include "kernels.h"

namespace mb {

constexpr int PITCH = 144;   // bytes per LDS row

template <class T, int BROWS, bool KMAJ>
struct Stager {
    static constexpr int EPV = 16 / sizeof(T);          // elements per 16-byte vector
    static constexpr int BKE = 128 / sizeof(T);         // k elements per tile
    static constexpr int NREG = BROWS / 8;              // staging dwords per thread
    uint32_t r[NREG];

    // base: operand pointer; ld: leading dimension (elements); row0: first tile row; nrows: valid rows
    // k0: first k of tile; kend: k limit (kmaj operands only; row operands need K % BKE == 0)
    __device__ __forceinline__ void load(const T* __restrict__ base, int ld, int row0, int nrows, int k0, int kend, int tid) {
        if constexpr (!KMAJ) {
#pragma unroll
            for (int i = 0; i < BROWS / 32; ++i) {
                const int rr = (tid >> 3) + 32 * i, c = tid & 7;
                const int g = row0 + rr;
                u32x4 v = {0u, 0u, 0u, 0u};
                if (g < nrows) v = *(const u32x4*)(base + (size_t)g * ld + k0 + c * EPV);
                r[4 * i + 0] = v[0]; r[4 * i + 1] = v[1]; r[4 * i + 2] = v[2]; r[4 * i + 3] = v[3];
            }
        } else {
            constexpr int DW = BROWS * (int)sizeof(T) / 4;      // dword columns per tile row
            constexpr int TASKS = 8 * DW / 256;
            constexpr int EPD = 4 / sizeof(T);                  // elements per dword
#pragma unroll
            for (int i = 0; i < TASKS; ++i) {
                const int id = i * 256 + tid;
                const int dwc = id % DW, kg = id / DW;
                const int col = row0 + dwc * EPD;
#pragma unroll
                for (int j = 0; j < EPV; ++j) {
                    const int kk = k0 + kg * EPV + j;
                    uint32_t v = 0u;
                    if (kk < kend && col < nrows) v = *(const uint32_t*)(base + (size_t)kk * ld + col);
                    r[i * EPV + j] = v;
                }
            }
        }
    }

    __device__ __forceinline__ void store(char* lds, int tid) const {
        if constexpr (!KMAJ) {
#pragma unroll
            for (int i = 0; i < BROWS / 32; ++i) {
                const int rr = (tid >> 3) + 32 * i, c = tid & 7;
                u32x4 v = {r[4 * i + 0], r[4 * i + 1], r[4 * i + 2], r[4 * i + 3]};
                *(u32x4*)(lds + rr * PITCH + c * 16) = v;
            }
        } else {
            constexpr int DW = BROWS * (int)sizeof(T) / 4;
            constexpr int TASKS = 8 * DW / 256;
#pragma unroll
            for (int i = 0; i < TASKS; ++i) {
                const int id = i * 256 + tid;
                const int dwc = id % DW, kg = id / DW;
                if constexpr (sizeof(T) == 2) {
                    u32x4 lo, hi;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const uint32_t a = r[i * 8 + 2 * q], b = r[i * 8 + 2 * q + 1];
                        lo[q] = (a & 0xFFFFu) | (b << 16);
                        hi[q] = (a >> 16) | (b & 0xFFFF0000u);
                    }
                    *(u32x4*)(lds + (2 * dwc) * PITCH + kg * 16) = lo;
                    *(u32x4*)(lds + (2 * dwc + 1) * PITCH + kg * 16) = hi;
                } else {
                    u32x4 v = {r[i * 4 + 0], r[i * 4 + 1], r[i * 4 + 2], r[i * 4 + 3]};
                    *(u32x4*)(lds + dwc * PITCH + kg * 16) = v;
                }
            }
        }
    }
};

template <class T, int BM, int BN, bool AK, bool BK, int MODE>
__global__ void __launch_bounds__(256) gemm_kernel(const GemmArgs p) {
    constexpr int BKE = 128 / sizeof(T);
    constexpr int MT = BM / 32, NT = BN / 32;
    typedef typename Frag<T>::type frag_t;
    __shared__ __attribute__((aligned(16))) char smem[2 * (BM + BN) * PITCH];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;

    // XCD-aware tile order: block b runs on XCD b % 8; give each XCD a contiguous run of tiles so that
    // tiles sharing an A row-panel / neighbouring B panels hit the same private L2 (bijective for any nwg).
    const int tiles_n = (p.N + BN - 1) / BN;
    const int nwg = gridDim.x;
    int swz;
    {
        const int q = nwg >> 3, rem = nwg & 7, xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        swz = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + j;
    }
    const int m0 = (swz / tiles_n) * BM, n0 = (swz % tiles_n) * BN;

    const int kbeg = blockIdx.y * p.kchunk;
    const int kend = min(p.K, kbeg + p.kchunk);
    const int nt = (kend - kbeg + BKE - 1) / BKE;

    const T* __restrict__ A = (const T*)p.A;
    const T* __restrict__ B = (const T*)p.B;

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    Stager<T, BM, AK> sa;
    Stager<T, BN, BK> sb;

    if (nt > 0) {
        sa.load(A, p.lda, m0, p.M, kbeg, kend, tid);
        sb.load(B, p.ldb, n0, p.N, kbeg, kend, tid);
        sa.store(smem, tid);
        sb.store(smem + BM * PITCH, tid);
    }
    __syncthreads();

    for (int t = 0; t < nt; ++t) {
        char* cur = smem + (t & 1) * (BM + BN) * PITCH;
        char* nxt = smem + ((t + 1) & 1) * (BM + BN) * PITCH;
        const bool more = (t + 1 < nt);
        if (more) {
            sa.load(A, p.lda, m0, p.M, kbeg + (t + 1) * BKE, kend, tid);
            sb.load(B, p.ldb, n0, p.N, kbeg + (t + 1) * BKE, kend, tid);
        }
        const char* As = cur + (wr * (BM / 2) + (lane & 15)) * PITCH + (lane >> 4) * 16;
        const char* Bs = cur + BM * PITCH + (wc * (BN / 2) + (lane & 15)) * PITCH + (lane >> 4) * 16;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            frag_t a[MT], b[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) a[i] = *(const frag_t*)(As + i * 16 * PITCH + s * 64);
#pragma unroll
            for (int j = 0; j < NT; ++j) b[j] = *(const frag_t*)(Bs + j * 16 * PITCH + s * 64);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) mma16(acc[i][j], b[j], a[i]);
        }
        if (more) {
            sa.store(nxt, tid);
            sb.store(nxt + BM * PITCH, tid);
        }
        __syncthreads();
    }

    // ------------------------------------------------------------------ epilogue
    // acc[i][j][r] = C[m][n + r],  m = m0 + wr*BM/2 + i*16 + (lane&15),  n = n0 + wc*BN/2 + j*16 + (lane>>4)*4
    T* __restrict__ C = (T*)p.C;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int m = m0 + wr * (BM / 2) + i * 16 + (lane & 15);
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int n = n0 + wc * (BN / 2) + j * 16 + (lane >> 4) * 4;
            if (n >= p.N) continue;
            f32x4 v = acc[i][j];
            if constexpr (MODE == EPI_BIAS || MODE == EPI_BIAS_F32) {
                v *= p.alpha;
                if (p.bias) v += *(const f32x4*)(p.bias + n);
                if constexpr (MODE == EPI_BIAS) store4(C + (size_t)m * p.ldc + n, v);
                else store4(p.Cf + (size_t)m * p.ldc + n, v);
            } else if constexpr (MODE == EPI_BIAS_GELU) {
                v += *(const f32x4*)(p.bias + n);
                f32x4 g = {gelu_f(v[0]), gelu_f(v[1]), gelu_f(v[2]), gelu_f(v[3])};
                store4(C + (size_t)m * p.ldc + n, v);
                store4((T*)p.C2 + (size_t)m * p.ldc + n, g);
            } else if constexpr (MODE == EPI_BIAS_DROP_RES) {
                v += *(const f32x4*)(p.bias + n);
                const uint32_t idx = (uint32_t)m * (uint32_t)p.N + (uint32_t)n;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] *= drop_mult(p.drop, idx + r);
                v += load4((const T*)p.R + (size_t)m * p.ldr + n);
                store4(C + (size_t)m * p.ldc + n, v);
            } else if constexpr (MODE == EPI_ADD_RES) {
                if (p.R) v += load4((const T*)p.R + (size_t)m * p.ldr + n);
                store4(C + (size_t)m * p.ldc + n, v);
            } else if constexpr (MODE == EPI_DGELU) {
                f32x4 u = load4((const T*)p.R + (size_t)m * p.ldr + n);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] *= dgelu_f(u[r]);
                store4(C + (size_t)m * p.ldc + n, v);
            } else if constexpr (MODE == EPI_ACCUM_F32) {
                float* dst = p.Cf + (size_t)m * p.ldc + n;
                if (gridDim.y > 1) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) atomicAdd(dst + r, v[r]);
                } else {
                    f32x4 o = *(f32x4*)dst;
                    o += v;
                    *(f32x4*)dst = o;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------- host
template <class T, int BM, int BN, bool AK, bool BK, int MODE>
static int launch_cfg(const GemmArgs& a, int splits, hipStream_t st) {
    GemmArgs p = a;
    constexpr int BKE = 128 / sizeof(T);
    const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    if (splits < 1) splits = 1;
    int kchunk = (p.K + splits - 1) / splits;
    kchunk = (kchunk + BKE - 1) / BKE * BKE;
    splits = (p.K + kchunk - 1) / kchunk;
    p.kchunk = kchunk;
    dim3 grid(tiles, splits);
    hipLaunchKernelGGL((gemm_kernel<T, BM, BN, AK, BK, MODE>), grid, dim3(256), 0, st, p);
    return (int)hipGetLastError();
}

template <class T, bool AK, bool BK, int MODE>
static int launch_tile(const GemmArgs& a, int splits, int tile, hipStream_t st) {
    if (tile == 0) {   // heuristic: fill >= ~1 wave of the 256 CUs
        const long t128 = (long)((a.M + 127) / 128) * ((a.N + 127) / 128) * (splits < 1 ? 1 : splits);
        tile = (t128 >= 224) ? 128 : 64;
    }
    if (tile == 128) return launch_cfg<T, 128, 128, AK, BK, MODE>(a, splits, st);
    return launch_cfg<T, 64, 64, AK, BK, MODE>(a, splits, st);
}

template <class T>
static int launch_T(const GemmArgs& a, int layout, int mode, int splits, int tile, hipStream_t st) {
    constexpr int BKE = 128 / sizeof(T);
    if (a.N % 4 != 0) return MB_ERR_SHAPE;
    if (layout == GEMM_NT) {
        if (a.K % BKE != 0) return MB_ERR_SHAPE;
        switch (mode) {
            case EPI_BIAS: return launch_tile<T, false, false, EPI_BIAS>(a, splits, tile, st);
            case EPI_BIAS_F32: return launch_tile<T, false, false, EPI_BIAS_F32>(a, splits, tile, st);
            case EPI_BIAS_GELU: return launch_tile<T, false, false, EPI_BIAS_GELU>(a, splits, tile, st);
            case EPI_BIAS_DROP_RES: return launch_tile<T, false, false, EPI_BIAS_DROP_RES>(a, splits, tile, st);
            case EPI_ADD_RES: return launch_tile<T, false, false, EPI_ADD_RES>(a, splits, tile, st);
            default: return MB_ERR_MODE;
        }
    } else if (layout == GEMM_NN) {        // A row, B kmaj  (dgrad)
        if (a.K % BKE != 0) return MB_ERR_SHAPE;
        if (sizeof(T) == 2 && (a.N % 2)) return MB_ERR_SHAPE;
        switch (mode) {
            case EPI_ADD_RES: return launch_tile<T, false, true, EPI_ADD_RES>(a, splits, tile, st);
            case EPI_DGELU: return launch_tile<T, false, true, EPI_DGELU>(a, splits, tile, st);
            case EPI_BIAS_F32: return launch_tile<T, false, true, EPI_BIAS_F32>(a, splits, tile, st);
            default: return MB_ERR_MODE;
        }
    } else if (layout == GEMM_TN) {        // A kmaj, B kmaj (wgrad)
        if (sizeof(T) == 2 && ((a.N % 2) || (a.M % 2))) return MB_ERR_SHAPE;
        switch (mode) {
            case EPI_ACCUM_F32: return launch_tile<T, true, true, EPI_ACCUM_F32>(a, splits, tile, st);
            default: return MB_ERR_MODE;
        }
    }
    return MB_ERR_MODE;
}

int gemm_launch(int dtype, int layout, int mode, const GemmArgs& a, int splits, int tile, hipStream_t st) {
    if (dtype == DT_BF16) return launch_T<bf16>(a, layout, mode, splits, tile, st);
    if (dtype == DT_F32) return launch_T<float>(a, layout, mode, splits, tile, st);
    return MB_ERR_DTYPE;
}

}  // namespace mb
