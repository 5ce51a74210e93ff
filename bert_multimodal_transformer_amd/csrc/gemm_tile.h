// Building blocks shared by the GEMM kernels of gemm.hip (4-wave tiles) and gemm_pp.hip (8-wave ping-pong tiles): XCD-aware tile
// placement, the LDS-staged row-major epilogue, the global -> LDS DMA images with their fragment readers, and the pinned-asm helpers
// of the hand-scheduled k loops.  Device code only; the launchers stay in the .hip files.
#pragma once
#include <cstdlib>
#include <algorithm>
#include <type_traits>
#include "kernels.h"

namespace mb {

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

// Tile of block blockIdx.x in a grouped launch: problem index g (wave-uniform) and the tile's origin; false = no tile (padding).
template <int BM, int BN>
__device__ __forceinline__ bool grouped_tile_origin(const GroupedGemmArgs& ga, int& g, int& m0, int& n0) {
    const int bid = (int)blockIdx.x - ga.ride.blocks;      // (rider workgroups come first; a multiple of 8, so bid & 7 is still the XCD)
    g = 0;
    if (ga.chunk > 0) {
        // XCD-compact placement across the WHOLE group: all tiles of all problems form one list in "strip" order (strips of
        // `reg_n` tiles across the longer side of a problem, row-major inside a strip); XCD x (= blockIdx % 8) owns the x-th run
        // of `chunk` consecutive tiles.  A run is ~one strip: ~(short side + strip width) operand panels per XCD instead of the
        // (rows + columns) of eight separate regions in EVERY problem -- the panels cross the fabric ~2x less often.
        const int xcd = bid & 7, j = bid >> 3;
        const int lin = xcd * ga.chunk + j;
        if (j >= ga.chunk || lin >= ga.first[ga.count]) return false;
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
        return true;
    }
#pragma unroll
    for (int i = 1; i < MB_MAX_GROUP; ++i)
        if (i < ga.count && bid >= ga.first[i]) g = i;
    g = __builtin_amdgcn_readfirstlane(g);
    return tile_origin<BM, BN>(ga.g[g], m0, n0, bid - ga.first[g]);
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
// NG > 1 (with KS; the eight-wave 128 x 64 tile of gemm_pp.hip): NG groups of four k-split waves, group g owns tile rows
// [g * BM / NG, (g + 1) * BM / NG) -- its four partial tiles are staged behind those of the groups before it.
template <class T, int BM, int BN, int MODE, bool KS, int NW = 4, int NG = 1>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& p, f32x4 (&acc)[KS ? BM / NG / 16 : BM / (8 * NW)][KS ? BN / 16 : BN / 32],
                                              int m0, int n0, int wave, int lane, char* smem_,
                                              const EpiPre<T, BM, BN, MODE, NW>& pre) {
    typedef EpiPre<T, BM, BN, MODE, NW> Pre;
    static_assert(NG == 1 || (KS && NW == 4 * NG), "groups are groups of four k-split waves");
    constexpr int WM = NW / 2;                      // waves along m (each wave: BM / WM rows x BN / 2 columns)
    constexpr int GR = BM / NG;                     // rows of a staged tile
    constexpr int MT = KS ? GR / 16 : BM / (16 * WM), NT = KS ? BN / 16 : BN / 32;
    constexpr int RBY = BN * 4;                     // staged row bytes (fp32)
    constexpr int REG = GR * RBY;                   // one staged tile
    constexpr int NSUM = KS ? 4 : 1;
    constexpr int TPR = Pre::TPR, RPP = Pre::RPP, NR = Pre::NR;
    const int wr = KS ? 0 : (wave >> 1), wc = KS ? 0 : (wave & 1);
    char* const smem = smem_;
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
                static_assert(NG == 1, "(the grouped form has no column-masked store)");
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
        // (a four-column fp32 row pass -- whole 512-byte row segments per store instruction instead of two half-line dwordx4 stores --
        //  was measured and changed nothing: the 128-KB tile per CU leaves at the rate the fabric takes 28 MB of writes from all
        //  CUs at once, ~8 us, whatever the store shape: profiles/r06_store_shape_ab.txt, MB_GEMM_DBG=32 = the eight-column form)
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
            const int rl = NG > 1 ? r % GR : r;                                      // row inside its group's staged tiles
            const char* sb = smem + (NG > 1 ? (r / GR) * (4 * REG) : 0) + rl * RBY;
            f32x4 a = *(const f32x4*)(sb + ((ch ^ (rl & 7)) << 4));
            f32x4 b = *(const f32x4*)(sb + (((ch + 1) ^ (rl & 7)) << 4));
#pragma unroll
            for (int w = 1; w < NSUM; ++w) {
                a += *(const f32x4*)(sb + w * REG + ((ch ^ (rl & 7)) << 4));
                b += *(const f32x4*)(sb + w * REG + (((ch + 1) ^ (rl & 7)) << 4));
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
#pragma clang fp contract(off)
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
        return dma_voff_blk(i * NW + wave, ld, row0, nrows, lane);
    }
    // the same for 1-KB piece `blk` of the stage image (any assignment of pieces to waves fills the same image)
    static __device__ __forceinline__ uint32_t dma_voff_blk(int blk, int ld, int row0, int nrows, int lane) {
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
        // the same read from addresses the caller computed ahead of time (addr[j] = base[j] + stage offset): no VALU next to the reads
        template <int R, int S, class FR> __device__ __forceinline__ void emit_at(FR (&dst)[NF], const uint32_t (&addr)[NB]) const {
            if constexpr (!KMAJ) {
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst[R].v) : "v"(addr[S]), "n"(R * 16 * KB) : "memory");
            } else {
                constexpr int F = R / 2, H = R % 2;
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst[F].h[H]) : "v"(addr[F]), "n"((S * 32 + H * 4) * RB) : "memory");
            }
        }
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

// host helpers defined in gemm.hip
int gemm_env_int(const char* name, int dflt);
void gemm_log(const void* fn, hipStream_t st, const GemmArgs* p, int count);
void gemm_log_ride(const AdamRide& r);                                  // MB_GEMM_LOG=1: the optimizer update a grouped launch carried
unsigned long long* gemm_trace_buffer(int blocks, hipStream_t st);     // null unless MB_GEMM_TRACE=1
int gemm_dbg_flags();                                                   // MB_GEMM_DBG
// 8-wave ping-pong kernels (gemm_pp.hip), bf16, 256 x 128 tiles.  MB_ERR_MODE: that (layout, epilogue) pair is not instantiated.
int gemm_pp_launch(bool ak, bool bk, int mode, const GemmArgs& p, dim3 grid, hipStream_t st);
int gemm_pn_launch(bool ak, bool bk, int mode, const GemmArgs& p, dim3 grid, hipStream_t st, bool tall = false);      // 128 x 64 tiles, 128 k per stage (tall: 256 x 64, 64 k)
int gemm_pn_ride_launch(const GemmArgs& p, const AdamRide& ride, dim3 grid, hipStream_t st, bool tall = false);         // dgrad (row, k-major, + residual) with rider workgroups in front
int gemm_pp_grouped_launch(const GroupedGemmArgs& ga, int grid, hipStream_t st);      // grid = tiles only: ga.ride.blocks are added
#define MB_GEMM_LAUNCH(KERN, grid, block, st, arg, plog, cnt) \
    do { gemm_log((const void*)(KERN), st, plog, cnt); hipLaunchKernelGGL((KERN), grid, block, 0, st, arg); } while (0)

}  // namespace mb
