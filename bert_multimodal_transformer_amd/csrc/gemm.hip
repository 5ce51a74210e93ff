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
#include "gemm_tile.h"
#include "adamw_dev.h"

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

// The same kernel with AdamW riders (kernels.h AdamRide) in front of its tiles: a 64 x 64 launch of 456 tiles leaves 312 of the 768 block
// slots (three 48-KB rings per CU) empty on CUs whose memory pipes are mostly idle -- the tiles are bound by LDS reads and MFMA issue, the
// update by HBM.  A symbol of its own: the plain kernel's name is what profiles and PMC tables are keyed by.
template <class T, int BM, int BN, bool AK, bool BK, int MODE, int NSTAGE, int KB>
__global__ void __launch_bounds__(256, BM == 64 ? 3 : 2) gemm2_ride_kernel(const GemmArgs p, const AdamRide ride) {
    // (three waves per SIMD = three blocks per CU, like the plain kernel: left alone the rider branch took the kernel to 200 registers,
    //  two blocks per CU, and the 456 tiles to a second round -- +3.7 us per launch whatever the riders did)
    __shared__ __attribute__((aligned(1024))) char smem[Gemm2Smem<BM, BN, NSTAGE, KB, false>::BYTES];
    // riders LAST in the grid (unlike the grouped launch, whose riders want whole CUs): the tiles are placed as in the plain launch, at most
    // two per CU, and the riders take what is left.  In front of the tiles, 128 rider workgroups that did almost nothing cost the launch
    // 3.3 us -- some CUs then hold three tiles (profiles/r06_wgrad_operand_touch.txt)
    const int tiles = (int)gridDim.x - ride.blocks;
    if ((int)blockIdx.x >= tiles) {
        adam_ride_block<256, 2>(ride, (int)blockIdx.x - tiles);
        return;
    }
    int m0, n0;
    if (!tile_origin<BM, BN>(p, m0, n0, (int)blockIdx.x)) return;
    gemm2_body<T, BM, BN, AK, BK, MODE, NSTAGE, KB, false, 4>(p, m0, n0, 0, smem);
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
    // rider: an optimizer update in a slot no tile needs (kernels.h AdamRide).  Only the layers' groups (128 x 128 and larger tiles) carry
    // them: the branch costs registers (the 64 x 64 kernels of MAG's group went from 48 - 72 to 196 - 200 with it)
    if constexpr (BM >= 128) {
        if ((int)blockIdx.x < ga.ride.blocks) {
            adam_ride_block<256, 3>(ga.ride, (int)blockIdx.x);
            return;
        }
    }
    int g, m0, n0;
    if (!grouped_tile_origin<BM, BN>(ga, g, m0, n0)) return;
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
int gemm_env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}
static int env_int(const char* name, int dflt) { return gemm_env_int(name, dflt); }
// MB_GEMM_LOG=1: one stderr line per GEMM launch -- kernel symbol (as a kernel trace prints it), number of problems, launch FLOPs and
// the first problem's M N K -- so that a profile's per-symbol durations can be priced against what the symbol actually computed
// (bench.py's in-run trace: a captured step logs each of its launches once)
static int g_gemm_log = -1;
void gemm_log(const void* fn, hipStream_t st, const GemmArgs* p, int count) {
    if (g_gemm_log < 0) g_gemm_log = env_int("MB_GEMM_LOG", 0);
    if (!g_gemm_log) return;
    double fl = 0.0;
    for (int i = 0; i < count; ++i) fl += 2.0 * (double)p[i].M * (double)p[i].N * (double)p[i].K;
    const char* name = hipKernelNameRefByPtr(fn, st);
    fprintf(stderr, "[magbert gemm] %s problems=%d flop=%.0f M=%d N=%d K=%d\n", name ? name : "?", count, fl, p[0].M, p[0].N, p[0].K);
}

void gemm_log_ride(const AdamRide& r) {
    if (g_gemm_log < 0) g_gemm_log = env_int("MB_GEMM_LOG", 0);
    if (g_gemm_log && r.blocks > 0 && r.n4 > 0) fprintf(stderr, "[magbert ride] params=%zu blocks=%d\n", r.n4 * 4, r.blocks);      // (not the touch-only riders)
}

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
unsigned long long* gemm_trace_buffer(int blocks, hipStream_t st) {
    if (g_trace_on == 0) return nullptr;
    return trace_buffer(blocks, st);
}
static int g_impl = -1, g_stages = -1, g_dbg = 0;      // MB_GEMM_IMPL: 0 auto, 1 = register-staged v1, 2 = LDS-DMA v2 ; MB_GEMM_STAGES: 2|3|4

int gemm_dbg_flags() {
    if (g_impl < 0) { g_impl = env_int("MB_GEMM_IMPL", 0); g_stages = env_int("MB_GEMM_STAGES", 0); g_dbg = env_int("MB_GEMM_DBG", 0); }
    return g_dbg;
}

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
        // 8-wave ping-pong kernel (gemm_pp.hip; bf16): anything it cannot take goes to the 128 x 128 configuration
        if constexpr (sizeof(T) == 2 && BN == 128) {
            if (v2ok && splits == 1 && p.K / BKE >= 3) {
                const int rc = gemm_pp_launch(AK, BK, MODE, p, grid, st);
                if (rc != MB_ERR_MODE) return rc;
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

// Tile code 12872: the 128 x 64 eight-wave ping-pong tile (gemm_pp.hip, bf16) -- the default of the narrow (N = 768) launches since round 6.
// -> the padded tile count of the launch, 0 = this problem runs the 64 x 64 configuration: a k range that is not whole 128-deep stages or
// shorter than three of them, a segmented B, split-K, and (unless the caller named the tile) more than 256 padded tiles -- one tile per
// CU: a second round costs more than the 64 x 64 kernel's three blocks per CU (T = 4096: 384 tiles) -- or fewer than 168: small problems
// keep the finer 64 x 64 grid.
static int g_tile_n768 = -1;       // MB_GEMM_TILE_N768: tile code for auto-selected narrow GEMMs (12872 | 64 | 12864 | 128)
static int tile_n768() {
    if (g_tile_n768 < 0) g_tile_n768 = env_int("MB_GEMM_TILE_N768", 12872);
    return g_tile_n768;
}
// `tall` in: 1 = only the 256 x 64 form (tile code 25672), 0 = only 128 x 64 (12872), -1 = whichever makes one round (the auto selection:
// 128 x 64 first; the tall form -- 64-deep stages, K >= 192 -- where that one would need a second round, MB_GEMM_PT=0 turns it off);
// out: which one it is.
static int pn_cfg(const GemmArgs& a, bool ak, bool bk, int splits, bool forced, GemmArgs& p, int* tall = nullptr) {
    const int want = tall ? *tall : 0;
    if (tall) *tall = 0;
    static int pn_max = -1, pt_on = -1;         // MB_GEMM_PN_MAX (measurement switch): the largest padded tile count the auto selection takes
    if (pn_max < 0) { pn_max = env_int("MB_GEMM_PN_MAX", 256); pt_on = env_int("MB_GEMM_PT", 1); }
    if (g_impl < 0) { g_impl = env_int("MB_GEMM_IMPL", 0); g_stages = env_int("MB_GEMM_STAGES", 0); g_dbg = env_int("MB_GEMM_DBG", 0); }
    const bool common = splits <= 1 && g_impl != 1 && g_stages <= 0 && a.bseg <= 0 && (a.lda % 8 == 0) && (a.ldb % 8 == 0) &&
                        (((uintptr_t)a.A | (uintptr_t)a.B) % 16 == 0) && (!bk || a.N % 64 == 0);
    if (!common) return 0;
    for (int form = 0; form < 2; ++form) {
        if ((want == 0 && form == 1) || (want == 1 && form == 0) || (form == 1 && want < 0 && !pt_on)) continue;
        p = a;
        const int tiles = form ? choose_regions<256, 64>(p) : choose_regions<128, 64>(p);
        const int ke = form ? 64 : 128, bm = form ? 256 : 128;
        bool ok = (p.K % ke == 0) && p.K / ke >= 3 && (forced || (tiles <= pn_max && tiles >= 168));
        if (ak) ok = ok && (p.M % bm == 0);
        if (!ok) continue;
        p.kchunk = p.K;
        p.dbg = g_dbg;
        p.trace = nullptr;
        if (tall) *tall = form;
        return tiles;
    }
    return 0;
}

// A dgrad launch (GEMM_NN, bf16) with riders, for the two configurations that leave block slots free at T = 2400:
//   EPI_ADD_RES, N = 768  : the 64 x 64 three-slot kernel, 456 tiles in 768 slots (three 48-KB rings per CU)
//   EPI_DGELU,  N = 3072  : the 128 x 128 two-slot kernel, 456 tiles in 512 slots -- 56 CUs hold ONE tile and are half idle throughout
// -> the padded tile count of the launch launch_tile / launch_cfg above would make of `a` (and the block slots per CU), 0 = another kernel.
static int nn_ride_cfg(int mode, const GemmArgs& a, GemmArgs& p, int* per_cu, int* pn = nullptr) {      // *pn: 0 four-wave kernels, 1 = 128 x 64 ping-pong, 2 = 256 x 64
    constexpr int BKE = 64, EPV = 8;
    p = a;
    if (pn) *pn = 0;
    if (g_impl < 0) { g_impl = env_int("MB_GEMM_IMPL", 0); g_stages = env_int("MB_GEMM_STAGES", 0); g_dbg = env_int("MB_GEMM_DBG", 0); }
    static int plain = -1;          // every selection switch of launch_tile / launch_cfg at its default (else: the plain launch, no riders)
    if (plain < 0)
        plain = (env_int("MB_GEMM_TRACE", 0) == 0 && env_int("MB_GEMM_64_STAGES", 3) == 3 &&
                 env_int("MB_GEMM_KSPLIT", 0) == 0 && (env_int("MB_GEMM_TILE_BIG", 0) & ~4) == 0) ? 1 : 0;
    if (g_impl == 1 || g_stages > 0 || !plain || a.bseg > 0) return 0;
    if (mode != EPI_ADD_RES && mode != EPI_DGELU) return 0;
    const long t128 = (long)((a.M + 127) / 128) * ((a.N + 127) / 128);
    const bool big = t128 >= 224;
    if (!big && tile_n768() == 12872) {
        // the 128 x 64 ping-pong tile: ONE tile per CU (144 KB of LDS), the riders are whole idle CUs like the grouped weight gradient's
        int tall = -1;
        const int tiles = mode == EPI_ADD_RES ? pn_cfg(a, false, true, 1, false, p, &tall) : 0;
        if (tiles > 0) { if (per_cu) *per_cu = 1; if (pn) *pn = 1 + tall; return tiles; }
        p = a;
    } else if (!big && tile_n768() != 64) return 0;
    if (big != (mode == EPI_DGELU)) return 0;
    const int bn = big ? 128 : 64;
    if (a.N % bn != 0 || a.N % 8 != 0 || a.ldc % 8 != 0) return 0;
    if ((a.K % BKE) || (a.lda % EPV) || (a.ldb % EPV) || (((uintptr_t)a.A | (uintptr_t)a.B) % 16) || a.K / BKE < 2) return 0;
    const int tiles = big ? choose_regions<128, 128>(p) : choose_regions<64, 64>(p);
    // (beyond 512 tiles the plain launch of the 64 x 64 shapes is another kernel -- two slots, five blocks per CU; the 128 x 128 kernel stays the
    //  same and simply takes more rounds: at T = 4096 its 768 tiles are one and a half -- riders in the slots its second round leaves free were
    //  measured: 3.7 M parameters stretch the launch from 33 to 47 us, the step gains nothing (4.72 vs 4.72 ms) and loses 0.4 % once the narrow
    //  dgrads carry riders of their own, profiles/r06_c5_dgelu_riders.txt, r06_pt_first_ab.txt; MB_ADAMW_RIDE_DGELU_ROUNDS=1 allows them)
    static int dgelu_rounds = -1;
    if (dgelu_rounds < 0) dgelu_rounds = env_int("MB_ADAMW_RIDE_DGELU_ROUNDS", 0);
    if (tiles > (big && dgelu_rounds ? 4096 : 512)) return 0;
    if (per_cu) *per_cu = big ? 2 : 3;
    p.kchunk = p.K;
    p.dbg = g_dbg;
    p.trace = nullptr;
    if (mode == EPI_DGELU && p.colsum == nullptr && p.Cf != nullptr) { p.colsum = p.Cf; p.Cf = nullptr; }      // (as engine_common.h gemm())
    return tiles;
}
int gemm_nn_ride_tiles(int dtype, int mode, const GemmArgs& a, int* per_cu) {
    GemmArgs p;
    return dtype == DT_BF16 ? nn_ride_cfg(mode, a, p, per_cu) : 0;
}
int gemm_nn_ride_launch(int dtype, int mode, const GemmArgs& a, const AdamRide& ride, hipStream_t st) {
    GemmArgs p;
    int pn = 0;
    const int tiles = dtype == DT_BF16 ? nn_ride_cfg(mode, a, p, nullptr, &pn) : 0;
    if (tiles <= 0 || ride.blocks <= 0 || (ride.blocks & 7)) return MB_ERR_MODE;
    if (pn) return gemm_pn_ride_launch(p, ride, dim3(tiles), st, pn == 2);
    gemm_log_ride(ride);
    if (mode == EPI_DGELU) {
        gemm_log((const void*)(gemm2_ride_kernel<bf16, 128, 128, false, true, EPI_DGELU, 2, 128>), st, &p, 1);
        hipLaunchKernelGGL((gemm2_ride_kernel<bf16, 128, 128, false, true, EPI_DGELU, 2, 128>), dim3(tiles + ride.blocks), dim3(256), 0, st, p, ride);
    } else {
        gemm_log((const void*)(gemm2_ride_kernel<bf16, 64, 64, false, true, EPI_ADD_RES, 3, 128>), st, &p, 1);
        hipLaunchKernelGGL((gemm2_ride_kernel<bf16, 64, 64, false, true, EPI_ADD_RES, 3, 128>), dim3(tiles + ride.blocks), dim3(256), 0, st, p, ride);
    }
    return (int)hipGetLastError();
}

template <class T, bool AK, bool BK, int MODE>
static int launch_pn(const GemmArgs& a, int splits, bool forced, hipStream_t st, int tall = -1) {
    if constexpr (sizeof(T) == 2) {
        GemmArgs p;
        const int tiles = pn_cfg(a, AK, BK, splits, forced, p, &tall);
        if (tiles > 0) {
            p.trace = (g_trace_on != 0) ? trace_buffer(tiles, st) : nullptr;
            const int rc = gemm_pn_launch(AK, BK, MODE, p, dim3(tiles), st, tall == 1);
            if (rc != MB_ERR_MODE) return rc;      // (a layout / epilogue pair that is not instantiated: 64 x 64)
        }
    }
    return launch_cfg<T, 64, 64, AK, BK, MODE>(a, splits, st);
}
static int g_tile_big = -1;        // MB_GEMM_TILE_BIG: 1 = 256 x 128 eight-wave tiles where they make ONE round on the chip (default 0: measured slower)

template <class T, bool AK, bool BK, int MODE>
static int launch_tile(const GemmArgs& a, int splits, int tile, hipStream_t st) {
    const bool forced = tile != 0;
    if (tile == 0) {   // heuristic: fill >= ~1 wave of the 256 CUs
        const long t128 = (long)((a.M + 127) / 128) * ((a.N + 127) / 128) * (splits < 1 ? 1 : splits);
        if (g_tile_big < 0) g_tile_big = env_int("MB_GEMM_TILE_BIG", 0);
        tile = (t128 >= 224) ? 128 : tile_n768();
        // one 256 x 128 tile per CU: taken when the whole output is a single, reasonably full round of the 256 CUs
        const long t256 = (long)((a.M + 255) / 256) * ((a.N + 127) / 128);
        // (2: also when it takes more than one round; 4: the forward launches only -- row x row operands --, the dgrad launches keep their riders)
        if (g_tile_big && (g_tile_big != 4 || (!AK && !BK)) && sizeof(T) == 2 && splits <= 1 && (t256 <= 256 || g_tile_big == 2) && t256 >= 168) tile = 256;
    }
    if constexpr (sizeof(T) == 2) {
        if (tile == 256) return launch_cfg<T, 256, 128, AK, BK, MODE>(a, splits, st);
    }
    if (tile == 256) tile = 128;
    if (tile == 128) return launch_cfg<T, 128, 128, AK, BK, MODE>(a, splits, st);
    if (tile == 12864) return launch_cfg<T, 128, 64, AK, BK, MODE>(a, splits, st);
    if (tile == 12872) return launch_pn<T, AK, BK, MODE>(a, splits, forced, st, forced ? 0 : -1);      // (falls back to 64 x 64 by itself)
    if (tile == 25672) return launch_pn<T, AK, BK, MODE>(a, splits, forced, st, 1);                    // the 256 x 64 form by name
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
static int launch_grouped(const GemmArgs* probs, int count, hipStream_t st, int stages, bool adam, const AdamRide* ride) {
    constexpr int BKE = 128 / sizeof(T);
    constexpr int EPV = 16 / sizeof(T);
    GroupedGemmArgs ga;
    ga.count = count;
    ga.ride = AdamRide{};
    if (ride && ride->blocks > 0 && ride->n4 > 0) {
        if (ride->blocks % 8 || adam || BM < 128) return MB_ERR_ARG;
        ga.ride = *ride;
    }
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
        unsigned long long* tr = trace_buffer(grid + ga.ride.blocks, st);
        for (int i = 0; i < count; ++i) ga.g[i].trace = tr;
    }
    if constexpr (BM == 256) {          // 8-wave ping-pong tiles (gemm_pp.hip): one 256 x 128 tile per CU
        if (adam) return MB_ERR_MODE;
        for (int i = 0; i < count; ++i) if (ga.g[i].K / BKE < 3) return MB_ERR_SHAPE;
        static int g_big = -1;           // MB_GROUP_BIG: 1 = one wave per SIMD (4 waves, 128 x 64 wave tiles), 2 = 8-wave ping-pong (gemm_pp.hip)
        if (g_big < 0) g_big = env_int("MB_GROUP_BIG", 2);
        if (g_big == 2) return gemm_pp_grouped_launch(ga, grid, st);
        gemm_log_ride(ga.ride);
        MB_GEMM_LAUNCH((gemm2_grouped_tn_kernel<T, BM, BN, 2, 128>), dim3(grid + ga.ride.blocks), dim3(256), st, ga, ga.g, count);
        return (int)hipGetLastError();
    } else {
    int gst = stages > 0 ? stages : g_gstages;       // the caller's choice for THIS group (MAG's small one: below) or the global switch
    if (sizeof(T) == 2 && gst == 2 && ga.g[0].K / BKE < 2) gst = 3;      // (all problems of a group share K) single-stage k range: plain loop
    // A group of few 64 x 64 tiles (MAG: 360 tiles for 512 slots, 8 MFMAs per wave and stage) is pure load latency: one k stage
    // costs one memory round trip divided by the stages in flight.  4 | 5 ring slots of 128-byte rows = 64 | 80 KB, still 2 blocks
    // per CU.  (The 128 x 128 groups measured slower with any deeper ring: 128 KB would leave one block per CU.)
    if (adam) {          // (experiment: the default 128 x 128 two-slot configuration only)
        if constexpr (BM == 128 && BN == 128) {
            MB_GEMM_LAUNCH((gemm2_grouped_tn_adam_kernel<T, BM, BN, 2, 128>), dim3(grid + ga.ride.blocks), dim3(256), st, ga, ga.g, count);
            return (int)hipGetLastError();
        }
        return MB_ERR_MODE;
    }
    if constexpr (BM == 64 && BN == 64) {
        if (gst == 4) { MB_GEMM_LAUNCH((gemm2_grouped_tn_kernel<T, BM, BN, 4, 128>), dim3(grid + ga.ride.blocks), dim3(256), st, ga, ga.g, count); return (int)hipGetLastError(); }
        if (gst == 5) { MB_GEMM_LAUNCH((gemm2_grouped_tn_kernel<T, BM, BN, 5, 128>), dim3(grid + ga.ride.blocks), dim3(256), st, ga, ga.g, count); return (int)hipGetLastError(); }
    }
    if (gst == 24) {
        MB_GEMM_LAUNCH((gemm2_grouped_tn_kernel<T, BM, BN, 4, 64>), dim3(grid + ga.ride.blocks), dim3(256), st, ga, ga.g, count);
    } else if (gst == 25) {
        MB_GEMM_LAUNCH((gemm2_grouped_tn_kernel<T, BM, BN, 5, 64>), dim3(grid + ga.ride.blocks), dim3(256), st, ga, ga.g, count);
    } else if (gst >= 3) {
        MB_GEMM_LAUNCH((gemm2_grouped_tn_kernel<T, BM, BN, 3, 128>), dim3(grid + ga.ride.blocks), dim3(256), st, ga, ga.g, count);
    } else {
        MB_GEMM_LAUNCH((gemm2_grouped_tn_kernel<T, BM, BN, 2, 128>), dim3(grid + ga.ride.blocks), dim3(256), st, ga, ga.g, count);
    }
    gemm_log_ride(ga.ride);
    return (int)hipGetLastError();
    }
}

int gemm_grouped_tn_ok(int dtype, const GemmArgs* probs, int count, int tile) {
    const int BKE = dtype == DT_BF16 ? 64 : 32, EPV = dtype == DT_BF16 ? 8 : 4;
    if (count < 1 || count > MB_MAX_GROUP || (tile != 64 && tile != 128 && tile != 256)) return 0;
    if (tile == 256 && dtype != DT_BF16) return 0;      // 256 = the 256 x 128 ping-pong tile (bf16, at least three k stages)
    const int tn = tile == 256 ? 128 : tile;
    for (int i = 0; i < count; ++i) {
        const GemmArgs& p = probs[i];
        if (tile == 256 && p.K / BKE < 3) return 0;
        if (p.K % BKE || p.M % tile || p.N % tn || p.lda % EPV || p.ldb % EPV || (p.cvalid <= 0 && p.ldc % 8) || p.cvalid > p.N ||
            (((uintptr_t)p.A | (uintptr_t)p.B) % 16))
            return 0;
    }
    return 1;
}

int gemm_grouped_tn_launch(int dtype, const GemmArgs* probs, int count, int tile, hipStream_t st, int stages, bool adam, const AdamRide* ride) {
    if (count < 1 || count > MB_MAX_GROUP) return MB_ERR_ARG;
    if (dtype == DT_BF16 && tile == 256) return launch_grouped<bf16, 256, 128>(probs, count, st, stages, adam, ride);
    if (dtype == DT_BF16) return tile == 128 ? launch_grouped<bf16, 128, 128>(probs, count, st, stages, adam, ride) : launch_grouped<bf16, 64, 64>(probs, count, st, stages, adam, ride);
    if (dtype == DT_F32) return tile == 128 ? launch_grouped<float, 128, 128>(probs, count, st, stages, adam, ride) : launch_grouped<float, 64, 64>(probs, count, st, stages, adam, ride);
    return MB_ERR_DTYPE;
}

int gemm_launch(int dtype, int layout, int mode, const GemmArgs& a, int splits, int tile, hipStream_t st) {
    if (dtype == DT_BF16) return launch_T<bf16>(a, layout, mode, splits, tile, st);
    if (dtype == DT_F32) return launch_T<float>(a, layout, mode, splits, tile, st);
    return MB_ERR_DTYPE;
}

}  // namespace mb
