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
#include <cstdlib>
#include <algorithm>
#include <type_traits>
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

// XCD-aware tile placement.  Block b runs on XCD b % 8 (8 XCDs, private 4 MB L2 each).  The tile grid is cut into
// reg_m x reg_n = 8 rectangular regions, one per XCD, so the A row-panels and B column-panels an XCD touches fit its
// L2 and are fetched from HBM / Infinity Cache once per XCD instead of once per tile.  The grid is padded to
// 8 * (tiles per region); blocks that fall outside the tile grid exit.
template <int BM, int BN>
__device__ __forceinline__ bool tile_origin(const GemmArgs& p, int& m0, int& n0, int bid = blockIdx.x) {
    const int xcd = bid & 7, j = bid >> 3;
    const int xm = xcd / p.reg_n, xn = xcd % p.reg_n;
    const int tm = xm * p.tpr_m + j / p.tpr_n, tn = xn * p.tpr_n + j % p.tpr_n;
    m0 = tm * BM;
    n0 = tn * BN;
    return m0 < p.M && n0 < p.N && (j / p.tpr_n) < p.tpr_m;
}

// ------------------------------------------------------------------ shared epilogue
// The MFMA accumulator layout gives a lane 4 columns of 16 different rows: stored directly that is 32-byte pieces of
// 16 cache lines per instruction (measured: the epilogue alone was 60 % of the FFN-1 GEMM).  Instead the tile is
// transposed through LDS (the operand ring is free by now): fp32 tile [BM][BN], 16-byte chunks XOR-swizzled by row,
// then every thread owns 8 consecutive columns of a row -> bias / residual / output accesses are 16-32-byte vectors
// and a wave writes whole 128-512-byte row segments.
template <class T> struct Vec8;
template <> struct Vec8<bf16> {
    static __device__ __forceinline__ void load(const bf16* p, float (&v)[8]) {
        const bf16x8 x = *(const bf16x8*)p;
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = (float)x[r];
    }
    static __device__ __forceinline__ void store(bf16* p, const float (&v)[8]) {
        bf16x8 x;
#pragma unroll
        for (int r = 0; r < 8; ++r) x[r] = (bf16)v[r];
        *(bf16x8*)p = x;
    }
};
template <> struct Vec8<float> {
    static __device__ __forceinline__ void load(const float* p, float (&v)[8]) {
        const f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) { v[r] = a[r]; v[4 + r] = b[r]; }
    }
    static __device__ __forceinline__ void store(float* p, const float (&v)[8]) {
        *(f32x4*)p = f32x4{v[0], v[1], v[2], v[3]};
        *(f32x4*)(p + 4) = f32x4{v[4], v[5], v[6], v[7]};
    }
};

// What an epilogue reads from global memory -- the residual / gelu' rows of this thread, the bias, the dropout key of a
// replayed step graph -- is requested BEFORE the k loop (EpiPre::fetch): issued inside the row pass, each of these loads was a
// full memory round trip on the critical path of a tile whose MFMA work is already over (measured per launch: 1.7 us for the
// residual epilogues, 6 us for the gelu' one, whose rows were read two at a time).  bf16 only: eight fp32 rows would be 64 VGPRs.
template <class T, int BM, int BN, int MODE, int NW>
struct EpiPre {
    static constexpr int TPR = BN / 8;                    // threads per row in the row-major pass
    static constexpr int RPP = NW * 64 / TPR;             // rows per pass
    static constexpr int NR = BM / RPP;                   // rows per thread
    static constexpr bool HAS_R = sizeof(T) == 2 && (MODE == EPI_BIAS_DROP_RES || MODE == EPI_ADD_RES || MODE == EPI_DGELU);
    bf16x8 r[HAS_R ? NR : 1];
    float bias8[8];
    DropKey key;
    int ldc, cvalid, overwrite;         // scalars of the row pass, read from the kernel arguments HERE (pinned): left to the compiler, one of
                                        // them ended up as an s_load inside the k loop, whose waits count scalar loads (tests/test_host_cpu.py)
    __device__ __forceinline__ void fetch(const GemmArgs& p, int m0, int n0, int tid) {
        ldc = p.ldc; cvalid = p.cvalid; overwrite = p.overwrite;
        asm volatile("" : "+s"(ldc), "+s"(cvalid), "+s"(overwrite));
        key = p.drop;
        key.resolve();
        const int n = n0 + (tid % TPR) * 8;
#pragma unroll
        for (int q = 0; q < 8; ++q) bias8[q] = 0.f;
        if constexpr (MODE == EPI_BIAS || MODE == EPI_BIAS_F32 || MODE == EPI_BIAS_GELU || MODE == EPI_BIAS_DROP_RES) {
            if (p.bias && n < p.N) Vec8<float>::load(p.bias + n, bias8);
        }
        if constexpr (HAS_R) {
#pragma unroll
            for (int it = 0; it < NR; ++it) {
                const int m = m0 + tid / TPR + it * RPP;
                r[it] = bf16x8{};
                if (p.R && m < p.M && n < p.N) r[it] = *(const bf16x8*)((const bf16*)p.R + (size_t)m * p.ldr + n);
            }
        }
    }
};

// KS (k-split waves): every wave holds a partial sum of the WHOLE tile (its quarter of every k-stage); the four partial tiles
// are staged side by side and added in the row-major pass.
template <class T, int BM, int BN, int MODE, bool KS, int NW = 4>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& p, f32x4 (&acc)[KS ? BM / 16 : BM / (8 * NW)][KS ? BN / 16 : BN / 32],
                                              int m0, int n0, int wave, int lane, char* smem,
                                              const EpiPre<T, BM, BN, MODE, NW>& pre) {
    typedef EpiPre<T, BM, BN, MODE, NW> Pre;
    constexpr int WM = NW / 2;                      // waves along m (each wave: BM / WM rows x BN / 2 columns)
    constexpr int MT = KS ? BM / 16 : BM / (16 * WM), NT = KS ? BN / 16 : BN / 32;
    constexpr int RBY = BN * 4;                     // staged row bytes (fp32)
    constexpr int REG = BM * RBY;                   // one staged tile
    constexpr int NSUM = KS ? 4 : 1;
    constexpr int TPR = Pre::TPR, RPP = Pre::RPP, NR = Pre::NR;
    const int wr = KS ? 0 : (wave >> 1), wc = KS ? 0 : (wave & 1);
    char* stage = smem + (KS ? wave * REG : 0);
    __syncthreads();                                // every wave is done with the operand stages
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int r = wr * (BM / WM) + i * 16 + (lane & 15);
            const int ch = (wc * (BN / 2) + j * 16 + (lane >> 4) * 4) >> 2;
            *(f32x4*)(stage + r * RBY + ((ch ^ (r & 7)) << 4)) = acc[i][j];
        }
    __syncthreads();
    const int tid = threadIdx.x;
    if constexpr (MODE == EPI_ACCUM_F32) {
        if (pre.cvalid > 0) {                       // (uniform) a lane per column: a wave stores whole row segments, dword by dword
            constexpr int RPP2 = NW * 64 / BN;      // rows per pass
            const int cc = tid % BN, nn = n0 + cc;
#pragma unroll 4
            for (int r = tid / BN; r < BM; r += RPP2) {
                const int m = m0 + r;
                if (m >= p.M || nn >= pre.cvalid) continue;
                const int o = r * RBY + (((cc >> 2) ^ (r & 7)) << 4) + (cc & 3) * 4;
                float v = *(const float*)(smem + o);
#pragma unroll
                for (int w = 1; w < NSUM; ++w) v += *(const float*)(smem + w * REG + o);
                float* dst = p.Cf + (size_t)m * pre.ldc + nn;
                if (gridDim.y > 1) atomicAdd(dst, v);
                else if (pre.overwrite) *dst = v;
                else *dst += v;
            }
            return;
        }
    }
    const int c = (tid % TPR) * 8;
    const int n = n0 + c;
    float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    AdamArgs adam = {};
    float adam_omb1 = 0.f, adam_omb2 = 0.f, adam_decay = 0.f;
    if constexpr (MODE == EPI_WGRAD_ADAM) {
        adam = *(const AdamArgs*)p.bias;          // this step's scalars (the step prologue wrote them)
        adam_omb1 = 1.0f - adam.beta1; adam_omb2 = 1.0f - adam.beta2; adam_decay = adam.lr * adam.weight_decay;
    }
    const float (&bias8)[8] = pre.bias8;
    T* __restrict__ C = (T*)p.C;
    const DropKey& dkey = pre.key;
#pragma unroll
    for (int it = 0; it < NR; ++it) {
        const int r = tid / TPR + it * RPP;
        const int m = m0 + r;
        if (m >= p.M || n >= p.N) continue;
        float v[8];
        {
            const int ch = c >> 2;
            f32x4 a = *(const f32x4*)(smem + r * RBY + ((ch ^ (r & 7)) << 4));
            f32x4 b = *(const f32x4*)(smem + r * RBY + (((ch + 1) ^ (r & 7)) << 4));
#pragma unroll
            for (int w = 1; w < NSUM; ++w) {
                a += *(const f32x4*)(smem + w * REG + r * RBY + ((ch ^ (r & 7)) << 4));
                b += *(const f32x4*)(smem + w * REG + r * RBY + (((ch + 1) ^ (r & 7)) << 4));
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) { v[q] = a[q]; v[4 + q] = b[q]; }
        }
        // the residual / gelu' row: prefetched (bf16) or read here (fp32)
        float res[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if constexpr (MODE == EPI_BIAS_DROP_RES || MODE == EPI_ADD_RES || MODE == EPI_DGELU) {
            if constexpr (Pre::HAS_R) {
#pragma unroll
                for (int q = 0; q < 8; ++q) res[q] = (float)pre.r[it][q];
            } else {
                if (p.R) Vec8<T>::load((const T*)p.R + (size_t)m * p.ldr + n, res);
            }
        }
        const size_t off = (size_t)m * pre.ldc + n;
        if constexpr (MODE == EPI_BIAS || MODE == EPI_BIAS_F32) {
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = v[q] * p.alpha + bias8[q];
            if constexpr (MODE == EPI_BIAS) Vec8<T>::store(C + off, v);
            else Vec8<float>::store(p.Cf + off, v);
        } else if constexpr (MODE == EPI_BIAS_GELU) {
            float g[8];
            const uint32_t gidx = (uint32_t)m * (uint32_t)p.N + (uint32_t)n;   // XLNet drops the activation (modeling_xlnet FF)
            // C keeps gelu'(u), not u: the backward (EPI_DGELU) only ever needs u through gelu', and here u is still the fp32
            // accumulator (bf16 mode: the derivative of the unrounded pre-activation; fp32 mode: bit-identical to computing it later)
#pragma unroll
            for (int q = 0; q < 8; q += 2) {
                const f32x2 u = {v[q] + bias8[q], v[q + 1] + bias8[q + 1]};
                f32x2 gg, dg;
                gelu_pair(u, gg, dg);
                g[q] = gg.x;
                g[q + 1] = gg.y;
                v[q] = dg.x;
                v[q + 1] = dg.y;
            }
            if (dkey.thresh != 0u) {               // activation dropout (MAG-XLNet only): one uniform branch per row
#pragma unroll
                for (int q = 0; q < 8; ++q) g[q] *= drop_mult(dkey, gidx + q);
            }
            Vec8<T>::store(C + off, v);
            Vec8<T>::store((T*)p.C2 + off, g);
        } else if constexpr (MODE == EPI_BIAS_DROP_RES) {
            const uint32_t idx = (uint32_t)m * (uint32_t)p.N + (uint32_t)n;
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = (v[q] + bias8[q]) * drop_mult(dkey, idx + q) + res[q];
            Vec8<T>::store(C + off, v);
        } else if constexpr (MODE == EPI_ADD_RES) {
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] += res[q];
            Vec8<T>::store(C + off, v);
        } else if constexpr (MODE == EPI_DGELU) {
            const uint32_t gidx = (uint32_t)m * (uint32_t)p.N + (uint32_t)n;
#pragma unroll
            for (int q = 0; q < 8; ++q) { v[q] *= res[q] * drop_mult(dkey, gidx + q); cs[q] += v[q]; }      // R = gelu'(u) saved by EPI_BIAS_GELU
            Vec8<T>::store(C + off, v);
        } else if constexpr (MODE == EPI_WGRAD_ADAM) {
            // the gradient never leaves the CU: HF-AdamW on this thread's eight parameters (adamw.hip: adam_update_store, same order of
            // operations -> the same bits as storing the gradient and sweeping it later)
            float* pp_ = (float*)p.C + off; float* pm_ = (float*)p.C2 + off; float* pv_ = (float*)const_cast<void*>(p.R) + off;
            float pp[8], mm[8], vv[8];
            Vec8<float>::load(pp_, pp); Vec8<float>::load(pm_, mm); Vec8<float>::load(pv_, vv);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float g = v[q] * adam.grad_scale;
                mm[q] = adam.beta1 * mm[q] + adam_omb1 * g;
                vv[q] = adam.beta2 * vv[q] + adam_omb2 * g * g;
                pp[q] -= adam.step_size * (mm[q] / (sqrtf(vv[q]) + adam.eps));
                if (adam_decay > 0.f) pp[q] -= adam_decay * pp[q];
            }
            Vec8<float>::store(pp_, pp); Vec8<float>::store(pm_, mm); Vec8<float>::store(pv_, vv);
            if (p.colsum) Vec8<bf16>::store((bf16*)p.colsum + off, pp);
        } else if constexpr (MODE == EPI_ACCUM_F32) {
            float* dst = p.Cf + off;
            if (gridDim.y > 1) {
#pragma unroll
                for (int q = 0; q < 8; ++q) atomicAdd(dst + q, v[q]);
            } else if (pre.overwrite) {                // the gradient buffer is known to hold zeros: no read-modify-write
                Vec8<float>::store(dst, v);
            } else {
                float o[8];
                Vec8<float>::load(dst, o);
#pragma unroll
                for (int q = 0; q < 8; ++q) o[q] += v[q];
                Vec8<float>::store(dst, o);
            }
        }
    }
    if constexpr (MODE == EPI_DGELU) {
        // fused bias gradient: this thread summed its rows; lanes that share the column group differ by TPR in lane id.  The
        // four waves' sums meet in LDS so that a block issues ONE atomic per column (measured: with one per wave the atomics
        // alone were 15 us of the 39 us dgrad-ffn2 launch -- 233 K atomics on 3072 addresses).
        if (p.colsum && !(p.dbg & 16)) {
            __syncthreads();                            // the row pass is done with the staged tile
            float* red = (float*)smem;                  // [NW waves][BN]
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                float s = cs[q];
#pragma unroll
                for (int o = TPR; o < 64; o <<= 1) s += __shfl_xor(s, o, 64);
                if (lane < TPR) red[wave * BN + lane * 8 + q] = s;
            }
            __syncthreads();
            if (tid < BN && n0 + tid < p.N) {
                float t = (red[tid] + red[BN + tid]) + (red[2 * BN + tid] + red[3 * BN + tid]);
                if constexpr (NW == 8) t += (red[4 * BN + tid] + red[5 * BN + tid]) + (red[6 * BN + tid] + red[7 * BN + tid]);
                grad_add(p.acc, p.colsum + n0 + tid, t);
            }
        }
    }
}

template <class T, int BM, int BN, bool AK, bool BK, int MODE>
__global__ void __launch_bounds__(256) gemm_kernel(const GemmArgs p) {
    constexpr int BKE = 128 / sizeof(T);
    constexpr int MT = BM / 32, NT = BN / 32;
    typedef typename Frag<T>::type frag_t;
    __shared__ __attribute__((aligned(16))) char smem[2 * (BM + BN) * PITCH];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;

    int m0, n0;
    if (!tile_origin<BM, BN>(p, m0, n0)) return;

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

    EpiPre<T, BM, BN, MODE, 4> pre;
    pre.fetch(p, m0, n0, threadIdx.x);
    gemm_epilogue<T, BM, BN, MODE, false>(p, acc, m0, n0, wave, lane, smem, pre);
}


// ============================================================================================== v2: LDS-DMA ring
// Same math, different data path: both operands go global -> LDS with global_load_lds (16 B per lane, no VGPR
// round trip) into an NSTAGE-deep ring, loads for tiles t+1 .. t+NSTAGE-2 stay in flight across the single
// s_barrier per k-tile (counted s_waitcnt vmcnt, never 0 in steady state).
//   row operands : image [row][128 B], 16-B chunks XOR-swizzled with (row & 7) on the SOURCE address and on the
//                  fragment read (the DMA destination is lane-linear) -> conflict-free ds_read_b128
//   kmaj operands: image [k][rows] exactly as stored in HBM (coalesced 256/512-B row segments); bf16 MFMA fragments
//                  are built with ds_read_b64_tr_b16 (hardware 4x16 transpose, two reads per fragment), fp32 with
//                  four ds_read_b32.  32-B blocks XOR-swizzled with the k row against bank conflicts.
// Requirements (else the v1 kernel above runs): K % BKE == 0 (callers zero-pad the token dimension of wgrad
// operands), kmaj operands have rows % tile == 0, row strides are 16-byte multiples.
typedef __attribute__((ext_vector_type(4))) short s16x4;
#define LDS_PTR(p) ((__attribute__((address_space(3))) char*)(p))

// XOR applied to the 16-byte chunk index of k-row k of a kmaj image (bit 0 stays clear: the 32-byte block a 16-lane group
// of ds_read_b64_tr_b16 reads stays contiguous).  A half-wave of the transpose read touches 8 k-rows -- k0 + {0,1,2,3} and
// k0 + 8 + {0,1,2,3} -- at the same columns, so the swizzle must send those 8 rows to the 8 different 32-byte windows of the
// 256-byte bank line: bits (k & 3, k >> 3 & 1) for 256-byte rows, (k >> 1 & 1, k >> 3 & 1) for 128-byte rows (two rows per
// bank line, k & 1 already separates them).  The round-1 form ignored k >> 3: rows k and k + 8 collided (2-way conflict on
// every transpose read = the 33-50 % conflict cycles of profiles/r01_gemm_pmc.md); MB_GEMM_DBG & 8 selects it for A/B runs.
template <class T, int RB> __device__ __forceinline__ int kswz(int k, bool r1 = false) {
    if (sizeof(T) != 2 || r1) return RB >= 256 ? ((k & 3) << 1) : (((k >> 1) & 1) << 1);
    return RB >= 256 ? (((k & 3) | (((k >> 3) & 1) << 2)) << 1) : ((((k >> 1) & 1) | (((k >> 3) & 1) << 1)) << 1);
}

// KB = bytes of k per stage row (128 or 64).  A smaller KB halves the stage, so twice as many stages (bytes in flight)
// fit next to the stage being multiplied -- the fill rate of a CU is latency x bytes-in-flight bound.
template <class T, int BROWS, bool KMAJ, int KB, int NW = 4>
struct Dma {
    static constexpr int EPV = 16 / sizeof(T);
    static constexpr int RB = BROWS * (int)sizeof(T);      // kmaj image row bytes
    static constexpr int CPR = KB / 16;                    // 16-B chunks per row-image row (8 or 4)
    static constexpr int RPI = 64 / CPR;                   // row-image rows per 1-KB DMA piece (8 or 16)
    static constexpr int NI = BROWS * KB / (1024 * NW);    // 1-KB pieces per wave per stage
    static_assert(NI * 1024 * NW == BROWS * KB, "a stage image is a whole number of 1-KB pieces per wave");
    typedef typename Frag<T>::type frag_t;

    // physical chunk of logical chunk lc in row r of the row image (conflict-free ds_read_b128)
    static __device__ __forceinline__ int rswz(int lc, int r) {
        return KB == 256 ? (lc ^ (r & 15)) : KB == 128 ? (lc ^ (r & 7)) : (lc ^ ((r >> 1) & 3));
    }

    static __device__ __forceinline__ void issue(const T* __restrict__ base, int ld, int row0, int nrows, size_t k0,
                                                 char* lds, int lane, int wave, bool r1) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int blk = i * NW + wave;
            const T* src;
            if constexpr (!KMAJ) {
                const int r = blk * RPI + lane / CPR;
                const int lc = rswz(lane % CPR, r);       // XOR swizzle is an involution: physical -> logical
                int g = row0 + r;
                g = g < nrows ? g : nrows - 1;          // rows past the edge: any valid row (their outputs are never stored)
                src = base + (size_t)g * ld + k0 + lc * EPV;
            } else {
                constexpr int RPK = 1024 / RB;
                const int kl = blk * RPK + (lane * 16) / RB;
                const int pc = ((lane * 16) % RB) >> 4;
                const int lc = pc ^ kswz<T, RB>(kl, r1);
                src = base + (size_t)(k0 + kl) * ld + row0 + lc * EPV;
            }
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)LDS_PTR(lds + blk * 1024), 16, 0, 0);
        }
    }

    // fragment for image rows rbase + (lane & 15), k-slab s (64 bytes of k)
    static __device__ __forceinline__ frag_t frag(const char* lds, int rbase, int s, int lane, bool r1) {
        if constexpr (!KMAJ) {
            const int r = rbase + (lane & 15);
            const int lc = s * 4 + (lane >> 4);
            return *(const frag_t*)(lds + r * KB + (rswz(lc, r) << 4));
        } else if constexpr (sizeof(T) == 2) {
            const int i = lane & 15;
            const int colb = (rbase + (i & 3) * 4) * 2;
            union { s16x4 h[2]; bf16x8 v; } u;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int k = s * 32 + (lane >> 4) * 8 + h * 4 + (i >> 2);
                const int pc = (colb >> 4) ^ kswz<T, RB>(k, r1);
                // inline asm, not __builtin_amdgcn_ds_read_tr16_b64_v4i16: hipcc (ROCm 7.2) orders the builtin behind EVERY pending
                // LDS-DMA of the wave (s_waitcnt vmcnt(0) in front of the first transpose read of a stage) -- the loads of stage
                // t + 1 issued a few instructions earlier were waited for before stage t was multiplied, i.e. no overlap of fill and
                // MFMA inside a block in any kernel with a k-major operand (dgrads, wgrads).  The asm is invisible to that pass; what
                // it must do itself is wait for the data (lds_fence below) before the first MFMA reads the registers.
                const uint32_t addr = (uint32_t)(size_t)LDS_PTR(lds + k * RB + (pc << 4) + (colb & 15));
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(u.h[h]) : "v"(addr) : "memory");
            }
            return u.v;
        } else {
            const int colb = (rbase + (lane & 15)) * 4;
            f32x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = s * 16 + (lane >> 4) * 4 + j;
                const int pc = (colb >> 4) ^ kswz<T, RB>(k, r1);
                v[j] = *(const float*)(lds + k * RB + (pc << 4) + (colb & 15));
            }
            return v;
        }
    }

    // ------------------------------------------------------------------ bf16 main loop (gemm2_body): what the phase and
    // per-iteration stamps of round 3 showed is that a wave never waits for its stage -- it is busy ISSUING: ~65 clocks per
    // global_load_lds (64-bit address arithmetic per piece and stage) and a full LDS round trip in front of every 16 MFMAs.
    //   * DMA by buffer_load ... lds: the lane's byte offset inside the operand is loop invariant (one VGPR per piece), the
    //     stage advance is ONE scalar offset: a piece is s_mov m0 + buffer_load, no vector arithmetic at all;
    //   * fragment reads as inline asm from loop-invariant base addresses + immediate offsets, all slabs of a stage issued
    //     before the DMA of the next stage (whose issue then hides their latency), counted out with s_waitcnt lgkmcnt.
    // byte offset of this lane's 16 bytes of piece i from the operand base, at k = 0
    static __device__ __forceinline__ uint32_t dma_voff(int i, int ld, int row0, int nrows, int lane, int wave) {
        const int blk = i * NW + wave;
        if constexpr (!KMAJ) {
            const int r = blk * RPI + lane / CPR;
            const int lc = rswz(lane % CPR, r);
            int g = row0 + r;
            g = g < nrows ? g : nrows - 1;
            return ((uint32_t)g * (uint32_t)ld + (uint32_t)(lc * EPV)) * (uint32_t)sizeof(T);
        } else {
            constexpr int RPK = 1024 / RB;
            const int kl = blk * RPK + (lane * 16) / RB;
            const int pc = ((lane * 16) % RB) >> 4;
            const int lc = pc ^ kswz<T, RB>(kl);
            return ((uint32_t)kl * (uint32_t)ld + (uint32_t)(row0 + lc * EPV)) * (uint32_t)sizeof(T);
        }
    }
    // bytes the operand advances per k element
    static __device__ __forceinline__ uint32_t k_stride_bytes(int ld) { return (KMAJ ? (uint32_t)ld : 1u) * (uint32_t)sizeof(T); }
    static __device__ __forceinline__ void issue_buf(__amdgpu_buffer_rsrc_t rs, const uint32_t (&voff)[NI], uint32_t soff, char* lds, int wave) {
#pragma unroll
        for (int i = 0; i < NI; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)LDS_PTR(lds + (i * NW + wave) * 1024), 16,
                                                     (int)voff[i], (int)soff, 0, 0);
    }
    // Fragment reader of one wave: NB loop-invariant LDS byte addresses (stage 0) -- row images: one per k-slab of the stage
    // (the XOR swizzle moves with the slab, the 16-row step of fragment i is an immediate); k-major images: one per fragment
    // (the swizzle moves with the column group, the k-slab and the two halves of the transpose read are immediates).
    template <int NF, int NSLAB> struct Reader {
        static constexpr int NB = KMAJ ? NF : NSLAB;
        uint32_t base[NB];
        // img: LDS address of the operand image in stage 0; rbase: first image row of the wave; slab0: first slab of the wave
        __device__ __forceinline__ void init(uint32_t img, int rbase, int slab0, int lane) {
            const int l15 = lane & 15, q = lane >> 4;
            if constexpr (!KMAJ) {
                const int r = rbase + l15;
#pragma unroll
                for (int s = 0; s < NSLAB; ++s) base[s] = img + (uint32_t)(r * KB + (rswz((slab0 + s) * 4 + q, r) << 4));
            } else {
                const int kl = q * 8 + (l15 >> 2);            // + slab * 32 + h * 4: neither term reaches the swizzle bits
#pragma unroll
                for (int j = 0; j < NF; ++j) {
                    const int colb = (rbase + j * 16 + (l15 & 3) * 4) * 2;
                    const int pc = (colb >> 4) ^ kswz<T, RB>(kl);
                    base[j] = img + (uint32_t)((slab0 * 32 + kl) * RB + (pc << 4) + (colb & 15));
                }
            }
        }
        // fragment f of slab s (relative to slab0) in the stage at byte offset `st`
        template <int F, int S> __device__ __forceinline__ bf16x8 read(uint32_t st) const {
            union { s16x4 h[2]; bf16x8 v; } u;
            if constexpr (!KMAJ) {
                const uint32_t a = base[S] + st;
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(u.v) : "v"(a), "n"(F * 16 * KB) : "memory");
            } else {
                const uint32_t a = base[F] + st;
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(u.h[0]) : "v"(a), "n"(S * 32 * RB) : "memory");
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(u.h[1]) : "v"(a), "n"((S * 32 + 4) * RB) : "memory");
            }
            return u.v;
        }
        static constexpr int READS_PER_FRAG = KMAJ ? 2 : 1;
        // read instruction R (of NF * READS_PER_FRAG) of slab S into the fragment set dst
        template <int R, int S, class FR> __device__ __forceinline__ void emit(FR (&dst)[NF], uint32_t st) const {
            if constexpr (!KMAJ) {
                const uint32_t a = base[S] + st;
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst[R].v) : "v"(a), "n"(R * 16 * KB) : "memory");
            } else {
                constexpr int F = R / 2, H = R % 2;
                const uint32_t a = base[F] + st;
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst[F].h[H]) : "v"(a), "n"((S * 32 + H * 4) * RB) : "memory");
            }
        }
    };
};

// -DMB_GEMM_LOOPTRACE (measurement builds only, scripts/exp/r3): wave 0 of every block keeps shader-clock stamps of the first
// kLtIters k-loop iterations in LDS (top of the iteration / stage landed / barrier passed / DMA issued / MFMAs issued) and copies
// them behind the five phase stamps of the block: kTraceStride u64 per block instead of 8.
#ifdef MB_GEMM_LOOPTRACE
constexpr int kLtIters = 24, kLtPoints = 5, kTraceStride = 128;
#else
constexpr int kTraceStride = 8;
#endif
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// All LDS reads issued so far (the asm transpose reads included, which the compiler's own s_waitcnt bookkeeping does not see) have
// returned; `touch` pins a fragment behind the wait (an MFMA consuming it cannot be scheduled in front of the s_waitcnt).
__device__ __forceinline__ void lds_wait_all() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// at most N LDS reads still outstanding (LDS returns in order; the counter has 4 bits).  Only meaningful while no scalar load is
// in flight (those return out of order): the k loop of gemm2_body holds none -- tests/test_host_cpu.py checks the ISA for that.
template <int N> __device__ __forceinline__ void lds_wait() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N < 15 ? N : 15) : "memory"); }
union FragU { bf16x8 v; s16x4 h[2]; };     // an MFMA operand fragment; a transpose read fills one half
// acc += X Y^T as a pinned instruction: volatile asm statements keep their order, which is what lets the k loop place the LDS
// reads and DMA issues of the NEXT slab between the MFMAs of this one (a builtin MFMA is free to move; hipcc put all 32 of a
// stage behind the last wait).  vDst == SrcC: consecutive accumulations into one tile need no wait states.
__device__ __forceinline__ void mma16_pinned(f32x4& acc, const bf16x8& x, const bf16x8& y) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(x), "v"(y));
}
template <int N, class F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}
template <class F> __device__ __forceinline__ void touch(F& f) { asm volatile("" : "+v"(f)); }

template <int BM, int BN, int NSTAGE, int KB, bool KS = false>
struct Gemm2Smem { static constexpr int STAGE = (BM + BN) * KB;
                   static constexpr int EPI = BM * BN * 4 * (KS ? 4 : 1);
                   static constexpr int BYTES = NSTAGE * STAGE > EPI ? NSTAGE * STAGE : EPI; };   // ring, reused by the epilogue tile(s)

// KS = k-split waves (KB = 256 only): instead of a quarter of the tile for every k, each of the four waves owns the WHOLE
// BM x BN tile for one 64-byte k-slab of every stage.  A 64 x 64 tile cut four ways leaves a wave 32 x 32: four fragment reads
// (1 KB of LDS traffic) per MFMA pair -- ~240 B/clk per CU at full MFMA rate, i.e. the LDS port itself.  With the whole tile per
// wave it is half that, there is one barrier per 256 bytes of k instead of per 128, and the tile count (the only way a
// [2432 x 768] output fills 256 CUs) stays the same.  The four partial tiles meet in the LDS-staged epilogue.
//
// NW = 8 (BM = 256 only, never with KS): eight waves as 4 x 2, each still a 64 x 64 quarter of a 128-row half -- the register
// picture of the 128 x 128 kernel, two waves per SIMD from ONE block.  One 256 x 128 tile per CU moves 25 % fewer operand bytes
// through the CU's 64 B/clk fill port than two co-resident 128 x 128 tiles, keeps 96 KB of loads in flight instead of 64 KB
// (3-deep ring of 48-KB stages) and turns a [2400 x 3072] output into 240 tiles: one balanced round on 256 CUs instead of 456
// tiles of which the slowest CU runs two.
template <class T, int BM, int BN, bool AK, bool BK, int MODE, int NSTAGE, int KB, bool KS = false, int NW = 4>
__device__ __forceinline__ void gemm2_body(const GemmArgs& p, const int m0, const int n0, const int ky, char* smem) {
    static_assert(!KS || KB == 256, "k-split waves: four 64-byte slabs per stage");
    static_assert(NW == 4 || (NW == 8 && !KS), "4 waves (2 x 2) or 8 waves (4 x 2)");
    constexpr int BKE = KB / sizeof(T);
    constexpr int WM = NW / 2;
    constexpr int MT = KS ? BM / 16 : BM / (16 * WM), NT = KS ? BN / 16 : BN / 32;
    constexpr int STAGE = (BM + BN) * KB;
    typedef typename Frag<T>::type frag_t;
    typedef Dma<T, BM, AK, KB, NW> DA;
    typedef Dma<T, BN, BK, KB, NW> DB;
    constexpr int G = DA::NI + DB::NI;            // DMA instructions per wave per stage

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = KS ? 0 : (wave >> 1), wc = KS ? 0 : (wave & 1);
    const int kbeg = ky * p.kchunk;
    const int kend = min(p.K, kbeg + p.kchunk);
    const int nt = (kend - kbeg) / BKE;
    const T* __restrict__ A = (const T*)p.A;
    const T* __restrict__ B = (const T*)p.B;
    // segmented B (GemmArgs::bseg): a k-major B moves its base to the segment of this tile's columns; a row-major B jumps by
    // seg_extra bytes whenever the k loop crosses into the next segment (every seg_stages stages)
    int n0b = n0;
    int seg_stages = 0;
    uint32_t seg_extra = 0;
    if (p.bseg > 0) {
        if constexpr (BK) { B += (size_t)(n0 / p.bseg) * p.bseg_stride; n0b = n0 % p.bseg; }
        else { seg_stages = p.bseg / BKE; seg_extra = (uint32_t)((p.bseg_stride - (size_t)p.bseg) * sizeof(T)); }
    }
    int seg_left = seg_stages;
    auto seg_step = [&](uint32_t& sob_) {            // after every stage's B offset advance
        if (seg_stages > 0 && --seg_left == 0) { sob_ += seg_extra; seg_left = seg_stages; }
    };

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // phase stamps of this block (0 entry, 1 first stage landed, 2 k loop done, 3 epilogue issued, 4 its stores completed)
    auto stamp = [&](int k) {
        if (p.trace && tid == 0) p.trace[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * kTraceStride + k] = wall_clock64();
    };
#ifdef MB_GEMM_LOOPTRACE
    __shared__ uint32_t lt[kLtIters * kLtPoints];
#define MB_LT(t, j) do { if (p.trace && (t) < kLtIters && tid == 0) lt[(t) * kLtPoints + (j)] = (uint32_t)__builtin_readcyclecounter(); } while (0)
#else
#define MB_LT(t, j) do { } while (0)
#endif
    stamp(0);
    EpiPre<T, BM, BN, MODE, NW> pre;
    pre.fetch(p, m0, n0, tid);                   // in flight under the whole k loop (oldest loads: counted out first by vmcnt)
    auto wait_stage = [&](int t) {               // stage t must have landed; up to NSTAGE-2 younger stages may stay in flight
        const int younger = min(NSTAGE - 2, nt - 1 - t);
        if (NSTAGE >= 5 && younger >= 3) wait_vmcnt<3 * G>();
        else if (NSTAGE >= 4 && younger >= 2) wait_vmcnt<2 * G>();
        else if (NSTAGE >= 3 && younger >= 1) wait_vmcnt<G>();
        else wait_vmcnt<0>();
    };
#ifdef MB_GEMM_PLAIN_LOOP      // A/B builds (scripts/build_variant.py): the un-pipelined bf16 loop below for two ring slots as well
    constexpr bool kPipelined = false;
#else
    constexpr bool kPipelined = true;
#endif
    if constexpr (sizeof(T) == 2 && NSTAGE == 2 && kPipelined) {
        // ------------------------------------------------------------------ bf16, two ring slots: software-pipelined slab stream
        // Round-3 stamps (profiles/r03_gemm_looptrace.txt): a wave never waits for its stage, it is busy issuing -- per iteration
        // ~600 clocks of DMA issue, an exposed LDS round trip per slab, and only then 16 MFMAs.  Here every MFMA is a pinned asm
        // statement and the work for the NEXT slab rides between them: the fragments of slab sigma + 1 are read into the second
        // register set while slab sigma is multiplied, and the stage two ahead is requested from the same gaps.  Because a stage
        // lives in registers while it is multiplied, its ring slot is free as soon as everybody has READ it: two slots carry
        // "being read" + "in flight".  One barrier per stage; every wait is a full drain (no counting next to scalar loads).
        //   two slabs per stage:  A: MFMA(t, 0) | reads(t, 1)          sync: stage t+1 landed, reads done, barrier
        //                         B: MFMA(t, 1) | reads(t+1, 0), DMA(t+2 -> slot of t)
        //   one slab per stage (k-split waves, 64-byte rows): every step is a B step, register sets alternate per stage
        constexpr int NSLAB = KS ? 1 : KB / 64;
        typedef typename DA::template Reader<MT, NSLAB> RA;
        typedef typename DB::template Reader<NT, NSLAB> RB_;
        constexpr int NRA = MT * RA::READS_PER_FRAG, NRD = NRA + NT * RB_::READS_PER_FRAG;      // LDS read instructions per slab
        constexpr int NM = MT * NT;
        RA ra;
        RB_ rb;
        const uint32_t lds0 = (uint32_t)(size_t)LDS_PTR(smem);
        ra.init(lds0, wr * (BM / WM), KS ? wave : 0, lane);
        rb.init(lds0 + BM * KB, wc * (BN / 2), KS ? wave : 0, lane);
        uint32_t va[DA::NI], vb[DB::NI];
#pragma unroll
        for (int i = 0; i < DA::NI; ++i) va[i] = DA::dma_voff(i, p.lda, m0, p.M, lane, wave);
#pragma unroll
        for (int i = 0; i < DB::NI; ++i) vb[i] = DB::dma_voff(i, p.ldb, n0b, p.N, lane, wave);
        const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, -1, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, -1, 0x00020000);
        const uint32_t ksa = DA::k_stride_bytes(p.lda) * BKE, ksb = DB::k_stride_bytes(p.ldb) * BKE;      // operand bytes per stage
        uint32_t soa = DA::k_stride_bytes(p.lda) * (uint32_t)kbeg, sob = DB::k_stride_bytes(p.ldb) * (uint32_t)kbeg;
#ifdef MB_GEMM_ABLATE
        const bool no_dma = (p.dbg & 1) != 0, no_reads = (p.dbg & 4) != 0, no_mfma = (p.dbg & 2) != 0;
#else
        constexpr bool no_dma = false, no_reads = false, no_mfma = false;
#endif
        // DMA piece I of the next stage in k order into the ring slot at byte offset `slot`
        auto dma_piece = [&](auto ic, uint32_t slot) {
            constexpr int I = decltype(ic)::value;
            if (no_dma) return;
            if constexpr (I < DA::NI)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsa, (__attribute__((address_space(3))) void*)LDS_PTR(smem + slot + (I * NW + wave) * 1024),
                                                         16, (int)va[I], (int)soa, 0, 0);
            else
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsb, (__attribute__((address_space(3))) void*)LDS_PTR(smem + slot + BM * KB + ((I - DA::NI) * NW + wave) * 1024),
                                                         16, (int)vb[I - DA::NI], (int)sob, 0, 0);
        };
        auto issue_stage = [&](uint32_t slot) {
            static_for<G>([&](auto ic) { dma_piece(ic, slot); });
            soa += ksa; sob += ksb;
            seg_step(sob);
        };
        FragU fa[2][MT], fb[2][NT];
        // read instruction R of slab S of the stage at `st` into register set BUF
        auto read_one = [&](auto rc, auto sc, auto bc, uint32_t st) {
            constexpr int R = decltype(rc)::value, S = decltype(sc)::value, BUF = decltype(bc)::value;
            if (no_reads) return;
            if constexpr (R < NRA) ra.template emit<R, S>(fa[BUF], st);
            else rb.template emit<R - NRA, S>(fb[BUF], st);
        };
        // one step of the stream: the MFMAs of register set BUF, with RD ? the reads of (stage at st_next, slab S_NEXT) into the
        // other set : nothing, and DMA ? the pieces of the next stage into slot_dma : nothing, spread over the gaps (reads first:
        // they are needed at the next step, the stage has a whole iteration to land)
        auto step = [&](auto bc, auto snc, auto rdc, auto dmac, uint32_t st_next, uint32_t slot_dma) {
            constexpr int BUF = decltype(bc)::value, S_NEXT = decltype(snc)::value;
            constexpr bool RD = decltype(rdc)::value, DMA = decltype(dmac)::value;
            constexpr int NF = (RD ? NRD : 0) + (DMA ? G : 0);
            static_for<NM>([&](auto mc) {
                constexpr int M = decltype(mc)::value;
                if (!no_mfma) mma16_pinned(acc[M / NT][M % NT], fb[BUF][M % NT].v, fa[BUF][M / NT].v);
                constexpr int f0 = M * NF / NM, f1 = (M + 1) * NF / NM;
                static_for<f1 - f0>([&](auto fc) {
                    constexpr int F = f0 + decltype(fc)::value;
                    if constexpr (RD && F < NRD) read_one(std::integral_constant<int, F>{}, snc, std::integral_constant<int, BUF ^ 1>{}, st_next);
                    else dma_piece(std::integral_constant<int, F - (RD ? NRD : 0)>{}, slot_dma);
                });
            });
            if constexpr (DMA) { soa += ksa; sob += ksb; seg_step(sob); }
        };
        typedef std::integral_constant<int, 0> I0;
        typedef std::integral_constant<int, 1> I1;
        typedef std::true_type Y;
        typedef std::false_type N_;
        issue_stage(0);
        issue_stage(STAGE);
        wait_vmcnt<G>();
        __builtin_amdgcn_s_barrier();
        stamp(1);
        static_for<NRD>([&](auto rc) { read_one(rc, I0{}, I0{}, 0u); });      // the only exposed fragment read of the tile
        // The loop body is ONE path (the accumulators and both fragment sets are loop-carried through tied asm operands: a
        // branch between step variants inside the loop costs a register copy of all of them per iteration); the last two
        // stages, which request / read nothing further, are peeled.  nt >= 2 (the launcher sends shorter k ranges elsewhere).
        if constexpr (NSLAB == 2) {
            auto stage = [&](int t, auto rdc, auto dmac, bool more) {
                const uint32_t cur = (t & 1) ? (uint32_t)STAGE : 0u, nxt = (uint32_t)STAGE - cur;
                MB_LT(t, 0);
                lds_wait_all();
                step(I0{}, I1{}, Y{}, N_{}, cur, 0u);                // A: MFMA(t, 0) | reads(t, 1)
                MB_LT(t, 1);
                if (more) wait_vmcnt<0>();                          // stage t + 1 has landed (the stage after it is not requested yet)
                lds_wait_all();
                MB_LT(t, 2);
                __builtin_amdgcn_s_barrier();                       // ... for everybody, and everybody has read stage t out of its slot
                MB_LT(t, 3);
                step(I1{}, I0{}, rdc, dmac, nxt, cur);               // B: MFMA(t, 1) | reads(t+1, 0), DMA(t+2)
                MB_LT(t, 4);
            };
            int t = 0;
            for (; t + 2 < nt; ++t) stage(t, Y{}, Y{}, true);
            stage(t, Y{}, N_{}, true);
            stage(t + 1, N_{}, N_{}, false);
        } else {
            auto one = [&](auto bc, int t, auto rdc, auto dmac, bool more) {
                constexpr int BUF = decltype(bc)::value;
                constexpr uint32_t cur = BUF ? (uint32_t)STAGE : 0u, nxt = (uint32_t)STAGE - cur;      // even stages live in slot 0
                MB_LT(t, 0);
                if (more) wait_vmcnt<0>();
                lds_wait_all();
                MB_LT(t, 2);
                __builtin_amdgcn_s_barrier();
                MB_LT(t, 3);
                step(bc, I0{}, rdc, dmac, nxt, cur);
                MB_LT(t, 4);
            };
            int t = 0;
            for (; t + 3 < nt; t += 2) {
                one(I0{}, t, Y{}, Y{}, true);
                one(I1{}, t + 1, Y{}, Y{}, true);
            }
            if (nt - t == 3) {
                one(I0{}, t, Y{}, Y{}, true);
                one(I1{}, t + 1, Y{}, N_{}, true);
                one(I0{}, t + 2, N_{}, N_{}, false);
            } else {
                one(I0{}, t, Y{}, N_{}, true);
                one(I1{}, t + 1, N_{}, N_{}, false);
            }
        }
        wait_vmcnt<0>();
    } else if constexpr (sizeof(T) == 2) {
        // ------------------------------------------------------------------ bf16: buffer-addressed DMA, asm fragment reads
        constexpr int NSLAB = KS ? 1 : KB / 64;              // 64-byte k-slabs of a stage this wave multiplies
        typedef typename DA::template Reader<MT, NSLAB> RA;
        typedef typename DB::template Reader<NT, NSLAB> RB_;
        constexpr int RPS = MT * RA::READS_PER_FRAG + NT * RB_::READS_PER_FRAG;      // LDS read instructions per slab
        RA ra;
        RB_ rb;
        const uint32_t lds0 = (uint32_t)(size_t)LDS_PTR(smem);
        ra.init(lds0, wr * (BM / WM), KS ? wave : 0, lane);
        rb.init(lds0 + BM * KB, wc * (BN / 2), KS ? wave : 0, lane);
        uint32_t va[DA::NI], vb[DB::NI];
#pragma unroll
        for (int i = 0; i < DA::NI; ++i) va[i] = DA::dma_voff(i, p.lda, m0, p.M, lane, wave);
#pragma unroll
        for (int i = 0; i < DB::NI; ++i) vb[i] = DB::dma_voff(i, p.ldb, n0b, p.N, lane, wave);
        const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, -1, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, -1, 0x00020000);
        const uint32_t ksa = DA::k_stride_bytes(p.lda) * BKE, ksb = DB::k_stride_bytes(p.ldb) * BKE;      // operand bytes per stage
        uint32_t soa = DA::k_stride_bytes(p.lda) * (uint32_t)kbeg, sob = DB::k_stride_bytes(p.ldb) * (uint32_t)kbeg;
        auto issue = [&](int slot) {                 // the next stage in k order goes to ring slot `slot`
#ifdef MB_GEMM_ABLATE
            if (p.dbg & 1) return;
#endif
            DA::issue_buf(rsa, va, soa, smem + slot * STAGE, wave);
            DB::issue_buf(rsb, vb, sob, smem + slot * STAGE + BM * KB, wave);
            soa += ksa; sob += ksb;
            seg_step(sob);
        };
#pragma unroll
        for (int s = 0; s < NSTAGE - 1; ++s)
            if (s < nt) issue(s);
        for (int t = 0; t < nt; ++t) {
            MB_LT(t, 0);
            wait_stage(t);
            MB_LT(t, 1);
            __builtin_amdgcn_s_barrier();            // everyone's piece of stage t landed; everyone left stage t-1
            MB_LT(t, 2);
            if (t == 0) stamp(1);
            const uint32_t st = (uint32_t)((t % NSTAGE) * STAGE);
            bf16x8 a[NSLAB][MT], b[NSLAB][NT];
#ifdef MB_GEMM_ABLATE
            const bool no_reads = (p.dbg & 4) != 0, no_mfma = (p.dbg & 2) != 0;
#else
            constexpr bool no_reads = false, no_mfma = false;
#endif
            // every fragment of the stage is requested first ...
            static_for<NSLAB>([&](auto sc) {
                constexpr int S = decltype(sc)::value;
                static_for<MT>([&](auto ic) { constexpr int I = decltype(ic)::value; if (!no_reads) a[S][I] = ra.template read<I, S>(st); else asm volatile("" : "=v"(a[S][I])); });
                static_for<NT>([&](auto jc) { constexpr int J = decltype(jc)::value; if (!no_reads) b[S][J] = rb.template read<J, S>(st); else asm volatile("" : "=v"(b[S][J])); });
            });
            // ... then the DMA of the stage NSTAGE-1 ahead (into the slot everyone left at the barrier): its issue hides the LDS latency
            if (t + NSTAGE - 1 < nt) issue((t + NSTAGE - 1) % NSTAGE);
            MB_LT(t, 3);
            static_for<NSLAB>([&](auto sc) {
                constexpr int S = decltype(sc)::value;
#ifdef MB_GEMM_LOOPTRACE
                lds_wait<0>();                        // (the stamps are scalar memory reads: no counting next to them)
#else
                lds_wait<(NSLAB - 1 - S) * RPS>();    // slab S has returned, the younger slabs may still be in flight
#endif
#pragma unroll
                for (int i = 0; i < MT; ++i) touch(a[S][i]);
#pragma unroll
                for (int j = 0; j < NT; ++j) touch(b[S][j]);
                if (!no_mfma) {
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int j = 0; j < NT; ++j) mma16(acc[i][j], b[S][j], a[S][i]);
                }
                __builtin_amdgcn_sched_barrier(0);    // the MFMAs of slab S stay in front of the wait for slab S + 1
            });
            MB_LT(t, 4);
        }
    } else {
        // ------------------------------------------------------------------ fp32 parity mode: global_load_lds + compiler-visible reads
        auto issue = [&](int t) {
            char* st = smem + (t % NSTAGE) * STAGE;
            DA::issue(A, p.lda, m0, p.M, kbeg + t * BKE, st, lane, wave, false);
            const int kb = kbeg + t * BKE;               // row-major segmented B: k -> (segment, k inside it)
            const size_t kB = seg_stages > 0 ? (size_t)(kb / p.bseg) * p.bseg_stride + (size_t)(kb % p.bseg) : (size_t)kb;
            DB::issue(B, p.ldb, n0b, p.N, kB, st + BM * KB, lane, wave, false);
        };
#pragma unroll
        for (int s = 0; s < NSTAGE - 1; ++s)
            if (s < nt) issue(s);
        for (int t = 0; t < nt; ++t) {
            wait_stage(t);
            __builtin_amdgcn_s_barrier();
            if (t == 0) stamp(1);
            if (t + NSTAGE - 1 < nt) issue(t + NSTAGE - 1);
            const char* cur = smem + (t % NSTAGE) * STAGE;
#pragma unroll
            for (int s0 = 0; s0 < (KS ? 1 : KB / 64); ++s0) {
                const int s = KS ? wave : s0;
                frag_t a[MT], b[NT];
#pragma unroll
                for (int i = 0; i < MT; ++i) a[i] = DA::frag(cur, wr * (BM / WM) + i * 16, s, lane, false);
#pragma unroll
                for (int j = 0; j < NT; ++j) b[j] = DB::frag(cur + BM * KB, wc * (BN / 2) + j * 16, s, lane, false);
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) mma16(acc[i][j], b[j], a[i]);
            }
        }
    }
    stamp(2);
    gemm_epilogue<T, BM, BN, MODE, KS, NW>(p, acc, m0, n0, wave, lane, smem, pre);
    if (p.trace) {
        stamp(3);
        wait_vmcnt<0>();
        stamp(4);
#ifdef MB_GEMM_LOOPTRACE
        if (tid == 0) {
            unsigned long long* dst = p.trace + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * kTraceStride + 8;
            for (int i = 0; i < kLtIters * kLtPoints; ++i) dst[i] = lt[i];
        }
#endif
    }
#undef MB_LT
}

template <class T, int BM, int BN, bool AK, bool BK, int MODE, int NSTAGE, int KB, bool KS = false, int NW = 4>
__global__ void __launch_bounds__(NW * 64) gemm2_kernel(const GemmArgs p) {
    __shared__ __attribute__((aligned(1024))) char smem[Gemm2Smem<BM, BN, NSTAGE, KB, KS>::BYTES];
    int m0, n0;
    if (!tile_origin<BM, BN>(p, m0, n0, blockIdx.x)) return;
    gemm2_body<T, BM, BN, AK, BK, MODE, NSTAGE, KB, KS, NW>(p, m0, n0, blockIdx.y, smem);
}

// Grouped wgrad: up to MB_MAX_GROUP independent dW += dY^T X problems in ONE launch.  Each of a layer's four weight
// gradients alone is at most ~2 blocks per CU (one under-filled round whose duration is set by the K = T loop latency,
// not by its size); launched together they are one grid of ~7 blocks per CU that keeps every CU's LDS ring full.
// Problem g owns blocks [first[g], first[g+1]) (multiples of 8, so block -> XCD mapping is unchanged).
template <class T, int BM, int BN, int NSTAGE, int KB, int MODE>
__device__ __forceinline__ void grouped_tn_block(const GroupedGemmArgs& ga) {
    __shared__ __attribute__((aligned(1024))) char smem[Gemm2Smem<BM, BN, NSTAGE, KB>::BYTES];
    // (A persistent variant that holds only one LDS slot per CU -- MB_GROUP_GRID = 128 / 256 / 384 blocks looping over the 432
    //  tiles -- was measured: no gain at 384, slower below; the launch is needed at full width to finish inside its layer.)
    int g = 0, m0, n0;
    if (ga.chunk > 0) {
        // XCD-compact placement across the WHOLE group: all tiles of all problems form one list in "strip" order (strips of
        // `reg_n` tiles across the longer side of a problem, row-major inside a strip); XCD x (= blockIdx % 8) owns the x-th run
        // of `chunk` consecutive tiles.  A run is ~one strip: ~(short side + strip width) operand panels per XCD instead of the
        // (rows + columns) of eight separate regions in EVERY problem -- the panels cross the fabric ~2x less often.
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        const int lin = xcd * ga.chunk + j;
        if (j >= ga.chunk || lin >= ga.first[ga.count]) return;
#pragma unroll
        for (int i = 1; i < MB_MAX_GROUP; ++i)
            if (i < ga.count && lin >= ga.first[i]) g = i;
        g = __builtin_amdgcn_readfirstlane(g);
        const GemmArgs& p = ga.g[g];
        const int local = lin - ga.first[g];
        const int W = p.reg_n, tm_n = p.tpr_m, tn_n = p.tpr_n;
        int tm, tn;
        if (p.reg_m == 0) {                          // strips across n
            const int strip = local / (tm_n * W), r = local - strip * tm_n * W;
            const int w = min(W, tn_n - strip * W);
            tm = r / w; tn = strip * W + r - tm * w;
        } else {                                     // strips across m
            const int strip = local / (tn_n * W), r = local - strip * tn_n * W;
            const int h = min(W, tm_n - strip * W);
            tn = r / h; tm = strip * W + r - tn * h;
        }
        m0 = tm * BM; n0 = tn * BN;
    } else {
#pragma unroll
        for (int i = 1; i < MB_MAX_GROUP; ++i)
            if (i < ga.count && (int)blockIdx.x >= ga.first[i]) g = i;
        g = __builtin_amdgcn_readfirstlane(g);
        if (!tile_origin<BM, BN>(ga.g[g], m0, n0, (int)blockIdx.x - ga.first[g])) return;
    }
    // (EPI_WGRAD_ADAM: touching the tile's p | m | v patch in front of the k loop -- one dword per 128-byte line, so that the epilogue
    //  finds it in the L2 / Infinity Cache -- was measured and lost: 91 instead of 80 us per launch, profiles/r05_adamw_in_wgrad_ab.txt)
    gemm2_body<T, BM, BN, true, true, MODE, NSTAGE, KB>(ga.g[g], m0, n0, 0, smem);
}
template <class T, int BM, int BN, int NSTAGE, int KB>
__global__ void __launch_bounds__(256) gemm2_grouped_tn_kernel(const GroupedGemmArgs ga) {
    grouped_tn_block<T, BM, BN, NSTAGE, KB, EPI_ACCUM_F32>(ga);
}
// (experiment, kernels.h EPI_WGRAD_ADAM: the same launch with HF-AdamW in the epilogue -- a kernel of its own so that the symbol of the
//  default one, which profiles and PMC tables are keyed by, stays what it was)
template <class T, int BM, int BN, int NSTAGE, int KB>
__global__ void __launch_bounds__(256) gemm2_grouped_tn_adam_kernel(const GroupedGemmArgs ga) {
    grouped_tn_block<T, BM, BN, NSTAGE, KB, EPI_WGRAD_ADAM>(ga);
}

// ---------------------------------------------------------------------------------------------- host
static int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}
// MB_GEMM_LOG=1: one stderr line per GEMM launch -- kernel symbol (as a kernel trace prints it), number of problems, launch FLOPs and
// the first problem's M N K -- so that a profile's per-symbol durations can be priced against what the symbol actually computed
// (bench.py's in-run trace: a captured step logs each of its launches once)
static int g_gemm_log = -1;
static void gemm_log(const void* fn, hipStream_t st, const GemmArgs* p, int count) {
    if (g_gemm_log < 0) g_gemm_log = env_int("MB_GEMM_LOG", 0);
    if (!g_gemm_log) return;
    double fl = 0.0;
    for (int i = 0; i < count; ++i) fl += 2.0 * (double)p[i].M * (double)p[i].N * (double)p[i].K;
    const char* name = hipKernelNameRefByPtr(fn, st);
    fprintf(stderr, "[magbert gemm] %s problems=%d flop=%.0f M=%d N=%d K=%d\n", name ? name : "?", count, fl, p[0].M, p[0].N, p[0].K);
}
#define MB_GEMM_LAUNCH(KERN, grid, block, st, arg, plog, cnt) \
    do { gemm_log((const void*)(KERN), st, plog, cnt); hipLaunchKernelGGL((KERN), grid, block, 0, st, arg); } while (0)

static unsigned long long* g_trace = nullptr;       // MB_GEMM_TRACE=1: device buffer of phase stamps, [kTraceBlocks][8]
static int g_trace_on = -1, g_trace_blocks = 0;
constexpr int kTraceBlocks = 8192 * 8 / kTraceStride;
static unsigned long long* trace_buffer(int blocks, hipStream_t st) {
    if (g_trace_on < 0) {
        const char* v = getenv("MB_GEMM_TRACE");
        g_trace_on = v ? atoi(v) : 0;
        if (g_trace_on && hipMalloc(&g_trace, (size_t)kTraceBlocks * kTraceStride * sizeof(unsigned long long)) != hipSuccess) g_trace_on = 0;
    }
    if (!g_trace_on || blocks > kTraceBlocks) return nullptr;
    g_trace_blocks = blocks;
    (void)hipMemsetAsync(g_trace, 0, (size_t)blocks * kTraceStride * sizeof(unsigned long long), st);
    return g_trace;
}
int gemm_trace_fetch(unsigned long long* host_out, int max_blocks) {
    if (!g_trace || !host_out) return 0;
    const int n = g_trace_blocks < max_blocks ? g_trace_blocks : max_blocks;
    if (hipDeviceSynchronize() != hipSuccess) return 0;
    if (hipMemcpy(host_out, g_trace, (size_t)n * kTraceStride * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return 0;
    return n;        // (a -DMB_GEMM_LOOPTRACE build hands out kTraceStride = 128 u64 per block: the caller sizes host_out for that)
}
static int g_impl = -1, g_stages = -1, g_dbg = 0;      // MB_GEMM_IMPL: 0 auto, 1 = register-staged v1, 2 = LDS-DMA v2 ; MB_GEMM_STAGES: 2|3|4

// choose the 8-region (one per XCD) decomposition with the smallest per-XCD panel footprint; returns the padded grid size
template <int BM, int BN>
static int choose_regions(GemmArgs& p) {
    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
    long best = -1;
    for (int rm = 1; rm <= 8; rm *= 2) {
        const int rn = 8 / rm;
        const int pm = (tiles_m + rm - 1) / rm, pn = (tiles_n + rn - 1) / rn;
        const long cost = (long)pm * BM + (long)pn * BN + 4L * ((long)pm * pn * 8 - (long)tiles_m * tiles_n);   // footprint + padding waste
        if (best < 0 || cost < best) { best = cost; p.reg_m = rm; p.reg_n = rn; p.tpr_m = pm; p.tpr_n = pn; }
    }
    return 8 * p.tpr_m * p.tpr_n;
}

template <class T, int BM, int BN, bool AK, bool BK, int MODE>
static int launch_cfg(const GemmArgs& a, int splits, hipStream_t st) {
    GemmArgs p = a;
    constexpr int BKE = 128 / sizeof(T);
    constexpr int EPV = 16 / sizeof(T);
    const int tiles = choose_regions<BM, BN>(p);
    if (splits < 1) splits = 1;
    int kchunk = (p.K + splits - 1) / splits;
    kchunk = (kchunk + BKE - 1) / BKE * BKE;
    splits = (p.K + kchunk - 1) / kchunk;
    p.kchunk = kchunk;
    dim3 grid(tiles, splits);
    if (g_impl < 0) { g_impl = env_int("MB_GEMM_IMPL", 0); g_stages = env_int("MB_GEMM_STAGES", 0); g_dbg = env_int("MB_GEMM_DBG", 0); }
    p.dbg = g_dbg;
    p.trace = (g_trace_on != 0) ? trace_buffer(tiles * splits, st) : nullptr;
    // v2 preconditions (see the kernel header)
    bool v2ok = (p.K % BKE == 0) && (p.lda % EPV == 0) && (p.ldb % EPV == 0) && (((uintptr_t)p.A | (uintptr_t)p.B) % 16 == 0);
    if (AK) v2ok = v2ok && (p.M % BM == 0);
    if (BK) v2ok = v2ok && (p.N % BN == 0);
    if (g_impl == 1) v2ok = false;
    if (p.bseg > 0) {          // segmented B: LDS-DMA kernels only, whole tiles / k-stages per segment, no split-K
        if (splits != 1 || (BK ? (p.bseg % BN != 0) : (p.bseg % 128 != 0)) || p.bseg_stride % EPV || !v2ok) return MB_ERR_SHAPE;
    }
    if constexpr (BM == 256) {
        // 8-wave single-round kernel (bf16, row-major A): anything it cannot take goes to the 128 x 128 configuration
        if constexpr (sizeof(T) == 2 && !AK && BN == 128) {
            if (v2ok && splits == 1) {
                MB_GEMM_LAUNCH((gemm2_kernel<T, 256, 128, AK, BK, MODE, 3, 128, false, 8>), grid, dim3(512), st, p, &p, 1);
                return (int)hipGetLastError();
            }
        }
        return launch_cfg<T, 128, 128, AK, BK, MODE>(a, splits, st);
    } else
    if (v2ok) {
        // (KB, NSTAGE) per tile: MB_GEMM_STAGES = 10*KBsel + stages overrides (KBsel 1 -> 128-byte rows, 2 -> 64-byte rows)
        int ns = 2, kb = 128;      // measured best (per-layer GEMM 290 us): deeper rings / 64-byte rows do not pay
        if (g_stages > 0) { kb = (g_stages / 10 == 2) ? 64 : 128; ns = g_stages % 10; }
        if (kb == 64 && (p.kchunk % (64 / (int)sizeof(T)) != 0)) kb = 128;
        // the two-slot bf16 loop is a software pipeline with its last two stages peeled: a k range of a single stage takes the
        // three-slot kernel (plain loop)
        if (sizeof(T) == 2 && ns <= 2 && p.kchunk / (kb / (int)sizeof(T)) < 2) ns = 3;
#define MB_LAUNCH2(NS, KBV) MB_GEMM_LAUNCH((gemm2_kernel<T, BM, BN, AK, BK, MODE, NS, KBV>), grid, dim3(256), st, p, &p, 1)
        static int g_ks = -1;             // MB_GEMM_KSPLIT: 1 = k-split waves for the 64 x 64 bf16 tiles (rounds 2-3), 0 (default) = quarter tiles
        if (g_ks < 0) g_ks = env_int("MB_GEMM_KSPLIT", 0);
        if constexpr (BM == 64 && BN == 64 && sizeof(T) == 2) {
            // each k-split block holds 64 KB of LDS (2 per CU): worth it while the whole grid is co-resident and the k loop is long
            // enough to amortise the four-tile epilogue (K = 768: 3 stages, measured 10.8 vs 9.8 us); beyond that
            // (T = 4096: 768 tiles) the 32 KB quarter-tile kernel's higher residency wins (measured: 7.30 vs 7.97 ms per step)
            if (g_ks && g_stages <= 0 && p.kchunk % 128 == 0 && p.kchunk >= 1024 && splits == 1 && tiles <= 512) {
                MB_GEMM_LAUNCH((gemm2_kernel<T, BM, BN, AK, BK, MODE, 2, 256, true>), grid, dim3(256), st, p, &p, 1);
                return (int)hipGetLastError();
            }
            // Round 4 (late): every 64 x 64 launch of at most 512 tiles takes the quarter-tile kernel with a THREE-slot ring of 128-byte
            // rows (48 KB: three blocks per CU, plain loop with two stages in flight) -- K = 768 (12 k stages, 8 MFMAs per wave and
            // stage: load latency, like MAG's weight gradients) AND the K >= 1024 problems the k-split kernel above used to take.
            // Same box, ms per step (profiles/r04_gemm64_ring_ab.txt): k-split + 2 slots 3.648 | k-split + 3 slots 3.62 | 3 slots
            // everywhere 3.60 | 4 slots everywhere 3.64; MAG-XLNet 4.25 -> 4.19.  Not beyond 512 tiles: at T = 4096 (768 tiles) the
            // two-slot kernel's five blocks per CU win (5.14 vs 5.19 ms).  MB_GEMM_64_STAGES=0 MB_GEMM_KSPLIT=1: the round-3 selection.
            static int g_64st = -1;
            if (g_64st < 0) g_64st = env_int("MB_GEMM_64_STAGES", 3);
            if (g_64st >= 3 && g_stages <= 0 && splits == 1 && p.kchunk / BKE >= 2 && tiles <= 512) ns = g_64st > 4 ? 4 : g_64st;
        }
        if (kb == 128) {
            if (BM == 128) { if (ns <= 2) MB_LAUNCH2(2, 128); else if (ns == 3) MB_LAUNCH2(3, 128); else MB_LAUNCH2(4, 128); }
            else { if (ns <= 2) MB_LAUNCH2(2, 128); else if (ns == 3) MB_LAUNCH2(3, 128); else MB_LAUNCH2(4, 128); }
        } else {
            if (ns <= 3) MB_LAUNCH2(3, 64); else if (ns == 4) MB_LAUNCH2(4, 64); else MB_LAUNCH2(5, 64);
        }
#undef MB_LAUNCH2
    } else {
        MB_GEMM_LAUNCH((gemm_kernel<T, BM, BN, AK, BK, MODE>), grid, dim3(256), st, p, &p, 1);
    }
    return (int)hipGetLastError();
}

static int g_tile_n768 = -1;       // MB_GEMM_TILE_N768: tile code for auto-selected narrow GEMMs (64 | 12864 | 128)
static int g_tile_big = -1;        // MB_GEMM_TILE_BIG: 1 = 256 x 128 eight-wave tiles where they make ONE round on the chip (default 0: measured slower)

template <class T, bool AK, bool BK, int MODE>
static int launch_tile(const GemmArgs& a, int splits, int tile, hipStream_t st) {
    if (tile == 0) {   // heuristic: fill >= ~1 wave of the 256 CUs
        const long t128 = (long)((a.M + 127) / 128) * ((a.N + 127) / 128) * (splits < 1 ? 1 : splits);
        if (g_tile_n768 < 0) g_tile_n768 = env_int("MB_GEMM_TILE_N768", 64);
        if (g_tile_big < 0) g_tile_big = env_int("MB_GEMM_TILE_BIG", 0);
        tile = (t128 >= 224) ? 128 : g_tile_n768;
        // one 256 x 128 tile per CU: taken when the whole output is a single, reasonably full round of the 256 CUs
        const long t256 = (long)((a.M + 255) / 256) * ((a.N + 127) / 128);
        if (g_tile_big && sizeof(T) == 2 && !AK && splits <= 1 && (t256 <= 256 || g_tile_big == 2) && t256 >= 168) tile = 256;      // (2: also when it takes more than one round)
    }
    if constexpr (sizeof(T) == 2 && !AK) {
        if (tile == 256) return launch_cfg<T, 256, 128, AK, BK, MODE>(a, splits, st);
    }
    if (tile == 256) tile = 128;
    if (tile == 128) return launch_cfg<T, 128, 128, AK, BK, MODE>(a, splits, st);
    if (tile == 12864) return launch_cfg<T, 128, 64, AK, BK, MODE>(a, splits, st);
    return launch_cfg<T, 64, 64, AK, BK, MODE>(a, splits, st);
}

template <class T>
static int launch_T(const GemmArgs& a, int layout, int mode, int splits, int tile, hipStream_t st) {
    constexpr int BKE = 128 / sizeof(T);
    if (a.N % 8 != 0 || a.ldc % 8 != 0) return MB_ERR_SHAPE;      // the row-major epilogue pass owns 8 columns per thread
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

template <class T, int BM, int BN>
static int launch_grouped(const GemmArgs* probs, int count, hipStream_t st, int stages, bool adam) {
    constexpr int BKE = 128 / sizeof(T);
    constexpr int EPV = 16 / sizeof(T);
    GroupedGemmArgs ga;
    ga.count = count;
    static int g_map = -1;           // MB_GROUP_MAP=0: round-1 placement (eight XCD regions inside every problem)
    if (g_map < 0) g_map = env_int("MB_GROUP_MAP", 1);
    int total = 0;
    for (int i = 0; i < count; ++i) {
        GemmArgs& p = ga.g[i];
        p = probs[i];
        // LDS-DMA path preconditions (as in launch_cfg): whole tiles, whole 128-byte K rows, 16-byte aligned operands
        if (p.K % BKE || p.M % BM || p.N % BN || p.lda % EPV || p.ldb % EPV || p.N % 8 || (p.cvalid <= 0 && p.ldc % 8) || p.cvalid > p.N ||
            (((uintptr_t)p.A | (uintptr_t)p.B) % 16))
            return MB_ERR_SHAPE;
        ga.first[i] = total;
        if (g_map) {
            p.tpr_m = p.M / BM; p.tpr_n = p.N / BN;
            p.reg_m = p.tpr_n >= p.tpr_m ? 0 : 1;               // strips run across the longer side
            total += p.tpr_m * p.tpr_n;
        } else {
            total += choose_regions<BM, BN>(p);
        }
        p.kchunk = p.K;
        if (g_impl < 0) { g_impl = env_int("MB_GEMM_IMPL", 0); g_stages = env_int("MB_GEMM_STAGES", 0); g_dbg = env_int("MB_GEMM_DBG", 0); }
        p.dbg = g_dbg;
    }
    ga.first[count] = total;
    static int g_gstages = -1;       // MB_GROUP_STAGES: ring of the grouped kernel: 2 | 3 stages of 128-byte k rows, 24 | 25 = 4 | 5 stages of 64-byte k rows
    if (g_gstages < 0) g_gstages = env_int("MB_GROUP_STAGES", 2);
    ga.chunk = g_map ? (total + 7) / 8 : 0;
    if (g_map)
        for (int i = 0; i < count; ++i) {       // strip width: one strip ~ one XCD's share (chunk) of the list
            const int shortside = std::min(ga.g[i].tpr_m, ga.g[i].tpr_n);
            ga.g[i].reg_n = std::max(1, (ga.chunk + shortside / 2) / shortside);
        }
    const int grid = g_map ? 8 * ga.chunk : total;
    if (g_trace_on != 0) {
        unsigned long long* tr = trace_buffer(grid, st);
        for (int i = 0; i < count; ++i) ga.g[i].trace = tr;
    }
    int gst = stages > 0 ? stages : g_gstages;       // the caller's choice for THIS group (MAG's small one: below) or the global switch
    if (sizeof(T) == 2 && gst == 2 && ga.g[0].K / BKE < 2) gst = 3;      // (all problems of a group share K) single-stage k range: plain loop
    // A group of few 64 x 64 tiles (MAG: 360 tiles for 512 slots, 8 MFMAs per wave and stage) is pure load latency: one k stage
    // costs one memory round trip divided by the stages in flight.  4 | 5 ring slots of 128-byte rows = 64 | 80 KB, still 2 blocks
    // per CU.  (The 128 x 128 groups measured slower with any deeper ring: 128 KB would leave one block per CU.)
    if (adam) {          // (experiment: the default 128 x 128 two-slot configuration only)
        if constexpr (BM == 128 && BN == 128) {
            MB_GEMM_LAUNCH((gemm2_grouped_tn_adam_kernel<T, BM, BN, 2, 128>), dim3(grid), dim3(256), st, ga, ga.g, count);
            return (int)hipGetLastError();
        }
        return MB_ERR_MODE;
    }
    if constexpr (BM == 64 && BN == 64) {
        if (gst == 4) { MB_GEMM_LAUNCH((gemm2_grouped_tn_kernel<T, BM, BN, 4, 128>), dim3(grid), dim3(256), st, ga, ga.g, count); return (int)hipGetLastError(); }
        if (gst == 5) { MB_GEMM_LAUNCH((gemm2_grouped_tn_kernel<T, BM, BN, 5, 128>), dim3(grid), dim3(256), st, ga, ga.g, count); return (int)hipGetLastError(); }
    }
    if (gst == 24) {
        MB_GEMM_LAUNCH((gemm2_grouped_tn_kernel<T, BM, BN, 4, 64>), dim3(grid), dim3(256), st, ga, ga.g, count);
    } else if (gst == 25) {
        MB_GEMM_LAUNCH((gemm2_grouped_tn_kernel<T, BM, BN, 5, 64>), dim3(grid), dim3(256), st, ga, ga.g, count);
    } else if (gst >= 3) {
        MB_GEMM_LAUNCH((gemm2_grouped_tn_kernel<T, BM, BN, 3, 128>), dim3(grid), dim3(256), st, ga, ga.g, count);
    } else {
        MB_GEMM_LAUNCH((gemm2_grouped_tn_kernel<T, BM, BN, 2, 128>), dim3(grid), dim3(256), st, ga, ga.g, count);
    }
    return (int)hipGetLastError();
}

int gemm_grouped_tn_ok(int dtype, const GemmArgs* probs, int count, int tile) {
    const int BKE = dtype == DT_BF16 ? 64 : 32, EPV = dtype == DT_BF16 ? 8 : 4;
    if (count < 1 || count > MB_MAX_GROUP || (tile != 64 && tile != 128)) return 0;
    for (int i = 0; i < count; ++i) {
        const GemmArgs& p = probs[i];
        if (p.K % BKE || p.M % tile || p.N % tile || p.lda % EPV || p.ldb % EPV || (p.cvalid <= 0 && p.ldc % 8) || p.cvalid > p.N ||
            (((uintptr_t)p.A | (uintptr_t)p.B) % 16))
            return 0;
    }
    return 1;
}

int gemm_grouped_tn_launch(int dtype, const GemmArgs* probs, int count, int tile, hipStream_t st, int stages, bool adam) {
    if (count < 1 || count > MB_MAX_GROUP) return MB_ERR_ARG;
    if (dtype == DT_BF16) return tile == 128 ? launch_grouped<bf16, 128, 128>(probs, count, st, stages, adam) : launch_grouped<bf16, 64, 64>(probs, count, st, stages, adam);
    if (dtype == DT_F32) return tile == 128 ? launch_grouped<float, 128, 128>(probs, count, st, stages, adam) : launch_grouped<float, 64, 64>(probs, count, st, stages, adam);
    return MB_ERR_DTYPE;
}

int gemm_launch(int dtype, int layout, int mode, const GemmArgs& a, int splits, int tile, hipStream_t st) {
    if (dtype == DT_BF16) return launch_T<bf16>(a, layout, mode, splits, tile, st);
    if (dtype == DT_F32) return launch_T<float>(a, layout, mode, splits, tile, st);
    return MB_ERR_DTYPE;
}

}  // namespace mb
