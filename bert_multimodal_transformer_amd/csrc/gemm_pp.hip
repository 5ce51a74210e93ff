// 256 x 128 bf16 GEMM tile run by EIGHT waves as two staggered four-wave groups ("ping-pong").
//
// Same math and the same callers as gemm.hip (the Linear layers under /root/reference/bert.py:221-229 and the weight gradients of
// loss.backward(), /root/reference/multimodal_driver.py:378); what differs is who is on the matrix pipe when.
//
// Why: two co-resident 128 x 128 blocks per CU need the CU's whole 64 B/clk L2 -> LDS fill path at full MFMA rate and sit at ~50 %
// of it.  One 256 x 128 tile moves 25 % fewer operand bytes per FLOP -- but eight waves behind ONE barrier issue DMA,
// read LDS and run MFMA in lockstep: every SIMD's matrix pipe idles while both of its waves read fragments (the round-2 kernel that
// lost).  Here the workgroup's two halves are offset by one barrier phase:
//
//     phase      2t                     2t+1                   2t+2
//     group 0    LOAD(t)                COMP(t)                LOAD(t+1)
//     group 1    COMP(t-1)              LOAD(t)                COMP(t)
//
//   LOAD(t): all fragment reads of k-stage t (64 registers: the wave's 64 x 64 quarter of its group's 128 x 128 half, two 32-deep
//            slabs) and NOTHING else; nothing for the matrix pipe
//   COMP(t): 32 MFMAs out of registers, with this wave's 6 DMA pieces of a later stage and the next LOAD's addresses between them
// A SIMD hosts wave w (group 0) and wave w+4 (group 1): in every phase one of them feeds the matrix pipe while the other feeds the
// LDS / DMA queues -- the complementary pairing MI355X_MICROARCH.md "Two waves per SIMD" item 5 asks for.  The two groups share the
// B image (128 columns) and the barrier; group 0 owns tile rows 0-127, group 1 rows 128-255.
//
// (That is the two-barrier form the kernel was built in, and the picture the phase names come from.  The default since late round 6 keeps
//  ONE of the two barriers per stage -- group 0 meets it behind COMP, group 1 behind LOAD; see MB_PP_ONE_BARRIER in the body -- with the
//  same per-wave instruction order, DMA schedule and waits.  gemm_pn_kernel is the same loop on a 128 x 64 tile with 128 k per stage.)
//
// Ring: three 48-KB slots.  In COMP(t) group 0 (phase 2t+1) requests its share of stage t+2, group 1 (phase 2t+2) its share of stage
// t+3 -- into the slot stage t-1 resp. t lived in, whose last reader drained its reads before the barrier in front of that phase
// (WAR).  Landing (RAW): a wave's pieces of stage t+1 are counted out (s_waitcnt vmcnt) in phase 2t+1 by BOTH groups -- group 0 at the
// end of COMP(t), group 1 at the end of LOAD(t) -- i.e. before the barrier in front of the first read of stage t+1 (group 0, phase
// 2t+2); every piece has had at least two phases to land by then.  Both groups run BOTH waits (no branch on the group in the loop):
// the other one is free for group 0 (nothing younger outstanding) and a phase early for group 1.
#include "gemm_tile.h"
#include "adamw_dev.h"

#ifndef MB_RIDE_UNR
#define MB_RIDE_UNR 3          // quads in flight per thread and pipeline stage of a rider workgroup (A/B builds: -DMB_RIDE_UNR=4|5|6)
#endif

namespace mb {

constexpr int kPpBM = 256, kPpBN = 128, kPpKB = 128, kPpSlots = 3, kPpWaves = 8;
constexpr int kPpStage = (kPpBM + kPpBN) * kPpKB;       // 48 KB
// The narrow form for the N = 768 launches (T = 2400: 228 tiles, one per CU, where 256 x 128 would make 60): 128 x 64 outputs, 128 k
// per stage (256-byte k rows) -- the SAME 48-KB stage, the same six DMA pieces per wave; a wave owns 32 x 32 outputs and four
// 32-deep slabs: 16 MFMAs and 16 (row image) / 24-32 (k-major) fragment reads per stage.  25 % fewer operand bytes per FLOP than
// the 64 x 64 tiles those launches run otherwise (they are bound by the L2 -> LDS fill).
constexpr int kPnBM = 128, kPnBN = 64, kPnKB = 256;
#ifndef MB_PN_KSW
#define MB_PN_KSW 0            // 1 = k-split waves in the narrow tile (measured: the loop gains 9 %, the four-partial epilogue takes it back)
#endif
constexpr bool kPnKsw = MB_PN_KSW != 0;
// The tall form for the same launches at T = 4096 (MOSEI shape: 128 x 64 would make 384 tiles = one and a half rounds): 256 x 64 outputs,
// 64 k per stage -- 40-KB stages, five DMA pieces per wave; a wave owns 64 x 32 outputs and two slabs (16 MFMAs, 12 / 16 reads per
// stage).  16 x 12 = 192 tiles at T = 4096: one round, and 64 CUs left to the riders.
constexpr int kPtBM = 256, kPtBN = 64, kPtKB = 128;
static_assert((kPnBM + kPnBN) * kPnKB == kPpStage, "both forms share the ring geometry");

#ifdef MB_GEMM_LOOPTRACE
// wave 0 (group 0) and wave 4 (group 1) stamp kPpIters k-stages from stage kPpFirst on, kPpPoints shader-clock stamps each:
// 0 top of LOAD | 1 reads issued | 2 reads returned (+ group 1: stage t+1 landed) | 3 barrier passed | 4 MFMAs + DMA issued |
// 5 group 0: stage t+1 landed  (the next stage's point 0 closes the second barrier)
constexpr int kPpFirst = 4, kPpIters = 10, kPpPoints = 6;
#endif

// dst = s + v as a pinned statement (stays between the MFMAs it is written between)
__device__ __forceinline__ void pinned_add(uint32_t& dst, uint32_t s, uint32_t v) { asm volatile("v_add_u32 %0, %1, %2" : "=v"(dst) : "s"(s), "v"(v)); }

// KSW (the narrow tile): the four waves of a group split the STAGE's k range instead of the group's outputs -- wave w multiplies slab w
// of every stage for the group's whole 64 x 64 half (16 accumulator tiles, partial sums added in the epilogue).  Half the fragment
// reads per MFMA of the 32 x 32-per-wave split: 8 (12 with a k-major B) instead of 16 (24) per stage, and both phases of the loop were
// as long as those reads take (profiles/r06_pn_looptrace.txt: COMP 464 clocks for 272 of MFMA next to a partner issuing 16 reads).
template <int BM, int BN, int KB, bool AK, bool BK, int MODE, bool KSW = false>
__device__ __forceinline__ void gemm_pp_body(const GemmArgs& p, const int m0, const int n0, char* smem) {
    typedef bf16 T;
    constexpr int NW = kPpWaves, STAGE = (BM + BN) * KB;
    constexpr int BKE = KB / 2;                    // 64 (128) k per stage
    static_assert(!KSW || BKE / 32 == 4, "k-split waves: one 32-deep slab per wave and stage");
    // a wave: 64 x 64 (32 x 32) outputs, two (four) 32-deep slabs per stage; KSW: the group's 64 x 64, one slab
    constexpr int MT = KSW ? BM / 32 : BM / 64, NT = KSW ? BN / 16 : BN / 32, NSLAB = KSW ? 1 : BKE / 32;
    typedef Dma<T, BM, AK, KB, NW> DA;
    typedef Dma<T, BN, BK, KB, NW> DB;
    constexpr int G = DA::NI + DB::NI;             // DMA pieces per wave per stage (4 + 2)
    typedef typename DA::template Reader<MT, NSLAB> RA;
    typedef typename DB::template Reader<NT, NSLAB> RB_;
    constexpr int NRA = MT * RA::READS_PER_FRAG, NRB = NT * RB_::READS_PER_FRAG;
    constexpr int NM = MT * NT * NSLAB;            // MFMAs per stage (32 | 16)
#ifdef MB_PP_SPREAD
    constexpr bool SPREAD = MB_PP_SPREAD != 0;     // (A/B builds)
#else
    constexpr bool SPREAD = NM < 32;               // DMA pieces at even distances over the MFMAs of COMP (else: behind the first MFMAs)
#endif

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;                     // waves w and w + 4 share a SIMD (MI355X_MICROARCH.md, LDS: dispatch order 0 -> 2 -> 1 -> 3)
    const int wr = wave >> 1, wc = wave & 1;       // 4 x 2 waves; wr 0-1 = group 0
    const int nt = p.K / BKE;                      // >= 3 (launcher)
    const T* __restrict__ A = (const T*)p.A;
    const T* __restrict__ B = (const T*)p.B;
    // segmented B (GemmArgs::bseg), as in gemm2_body
    int n0b = n0, seg_stages = 0;
    uint32_t seg_extra = 0;
    if (p.bseg > 0) {
        if constexpr (BK) { B += (size_t)(n0 / p.bseg) * p.bseg_stride; n0b = n0 % p.bseg; }
        else { seg_stages = p.bseg / BKE; seg_extra = (uint32_t)((p.bseg_stride - (size_t)p.bseg) * sizeof(T)); }
    }
    int seg_left = seg_stages;

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // The narrow tile has four accumulator tiles per wave: slab after slab into the same four, an MFMA meets its predecessor on that
    // tile four issues later.  NACC > 1: the slabs of a stage accumulate into NACC separate sets (summed behind the k loop), 16
    // independent chains like the 256 x 128 form.
#ifdef MB_PN_NACC
    constexpr int NACC = MB_PN_NACC;
#else
    constexpr int NACC = 1;        // (measured with 4 sets in the narrow tile: the same loop, profiles/r06_pn_nacc.txt -- the chain is not what COMP waits for)
#endif
    static_assert(NSLAB % NACC == 0, "whole slabs per accumulator set");
    f32x4 accx[NACC > 1 ? NACC - 1 : 1][MT][NT];
#pragma unroll
    for (int a = 0; a < (NACC > 1 ? NACC - 1 : 1); ++a)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) accx[a][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto stamp = [&](int k) {
        if (p.trace && tid == 0) p.trace[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * kTraceStride + k] = wall_clock64();
    };
#ifdef MB_GEMM_LOOPTRACE
    __shared__ uint32_t lt[2][kPpIters * kPpPoints];
#define PP_LT(t, j) do { if (p.trace && (t) >= kPpFirst && (t) < kPpFirst + kPpIters && (tid & 255) == 0) \
                             lt[grp][((t) - kPpFirst) * kPpPoints + (j)] = (uint32_t)__builtin_readcyclecounter(); } while (0)
#else
#define PP_LT(t, j) do { } while (0)
#endif
    stamp(0);
    EpiPre<T, BM, BN, MODE, NW> pre;
    pre.fetch(p, m0, n0, tid);                     // the oldest loads in flight: counted out by the first vmcnt wait

    RA ra;
    RB_ rb;
    const uint32_t lds0 = (uint32_t)(size_t)LDS_PTR(smem);
    if constexpr (KSW) {
        ra.init(lds0, grp * (BM / 2), wave & 3, lane);
        rb.init(lds0 + BM * KB, 0, wave & 3, lane);
    } else {
        ra.init(lds0, wr * (BM / 4), 0, lane);
        rb.init(lds0 + BM * KB, wc * (BN / 2), 0, lane);
    }
    // A wave's pieces of a stage are CONSECUTIVE 1-KB pieces of the image (wave * NI + i): one m0 per operand and stage, piece i is
    // the instruction's immediate offset i * 1024 -- which the hardware adds to the LDS address AND to the memory address, so the
    // lane offsets are taken back by i * 1024 and the descriptors' bases by kBias to keep them non-negative.
    constexpr uint32_t kBias = 4096;
    uint32_t va[DA::NI], vb[DB::NI];
#pragma unroll
    for (int i = 0; i < DA::NI; ++i) va[i] = DA::dma_voff_blk(wave * DA::NI + i, p.lda, m0, p.M, lane) + kBias - (uint32_t)i * 1024u;
#pragma unroll
    for (int i = 0; i < DB::NI; ++i) vb[i] = DB::dma_voff_blk(wave * DB::NI + i, p.ldb, n0b, p.N, lane) + kBias - (uint32_t)i * 1024u;
    const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)A - kBias), 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)B - kBias), 0, -1, 0x00020000);
    const uint32_t ksa = DA::k_stride_bytes(p.lda) * BKE, ksb = DB::k_stride_bytes(p.ldb) * BKE;      // operand bytes per stage
    // MB_PP_SPLIT_DMA=1 (experiment): the DMA pieces of a stage are issued in two halves: pieces [0, GA) behind the fragment reads of a
    // LOAD (the wave waits for LDS there anyway), pieces [GA, G) between the MFMAs of a COMP.  The narrow tile's COMP takes ~440 clocks
    // for 272 of MFMA whatever sits next to them (six pieces or three, spread or not, one accumulator chain or four, 16 partner reads
    // or 8: profiles/r06_pn_looptrace.txt, r06_pn_nacc.txt, r06_pn_ksw.txt, r06_pp_split_dma.txt).  GA <= DA::NI: the first half is operand
    // A only.  soa_a / dslot_a belong to the first half, soa_b / sob / dslot to the second (group 1's second half runs a stage ahead).
#ifndef MB_PP_ONE_BARRIER
#define MB_PP_ONE_BARRIER 1
#endif
#ifndef MB_PP_SPLIT_DMA
#define MB_PP_SPLIT_DMA 0      // measured (profiles/r06_pp_split_dma.txt): stand-alone -0.5 .. -1 us per launch, in the step nothing (+0.2 %); off
#endif
    constexpr bool ONEB = MB_PP_ONE_BARRIER != 0;   // one barrier per k-stage (below)
    constexpr int GA = MB_PP_SPLIT_DMA ? G / 2 : 0;
    static_assert(GA == 0 || ONEB, "the split DMA issue is written for the one-barrier loop");
    static_assert(GA <= DA::NI, "the LOAD half holds pieces of operand A only");
    uint32_t soa_a = 0, soa_b = 0, sob = 0;
#ifdef MB_GEMM_ABLATE
    const bool no_dma = (p.dbg & 1) != 0, no_reads = (p.dbg & 4) != 0, no_mfma = (p.dbg & 2) != 0;
#else
    constexpr bool no_dma = false, no_reads = false, no_mfma = false;
#endif
    // piece I (A pieces first) of this wave's share of the next stage in k order -> ring slot at byte offset `slot`
    auto dma_piece = [&](auto ic, uint32_t slot) {
        constexpr int I = decltype(ic)::value;
        if (no_dma) return;
        // (the immediate must be a literal: a template-dependent argument of the builtin fails the host pass)
#define MB_PP_PIECE(RS, PTR, VOFF, SOFF, J) do { \
            if constexpr ((J) == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(RS, PTR, 16, VOFF, SOFF, 0, 0); \
            else if constexpr ((J) == 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(RS, PTR, 16, VOFF, SOFF, 1024, 0); \
            else if constexpr ((J) == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(RS, PTR, 16, VOFF, SOFF, 2048, 0); \
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(RS, PTR, 16, VOFF, SOFF, 3072, 0); } while (0)
        static_assert(DA::NI <= 4 && DB::NI <= 4, "immediate offsets reach 4095");
        if constexpr (I < DA::NI)
            MB_PP_PIECE(rsa, (__attribute__((address_space(3))) void*)LDS_PTR(smem + slot + wave * (DA::NI * 1024)), (int)va[I], (int)(I < GA ? soa_a : soa_b), I);
        else
            MB_PP_PIECE(rsb, (__attribute__((address_space(3))) void*)LDS_PTR(smem + slot + BM * KB + wave * (DB::NI * 1024)), (int)vb[I - DA::NI], (int)sob, I - DA::NI);
#undef MB_PP_PIECE
    };
    auto dma_advance_a = [&]() { soa_a += ksa; };
    auto dma_advance = [&]() {                       // (the COMP half)
        soa_b += ksa; sob += ksb;
        if (seg_stages > 0 && --seg_left == 0) { sob += seg_extra; seg_left = seg_stages; }
    };
    auto issue_half_a = [&](uint32_t slot) {
        static_for<GA>([&](auto ic) { dma_piece(ic, slot); });
        dma_advance_a();
    };
    auto issue_half_b = [&](uint32_t slot) {
        static_for<G - GA>([&](auto ic) { dma_piece(std::integral_constant<int, GA + decltype(ic)::value>{}, slot); });
        dma_advance();
    };
    auto issue_stage = [&](uint32_t slot) { issue_half_a(slot); issue_half_b(slot); };
    // An instruction of the wave that is NOT on the matrix pipe costs a whole MFMA time (~16 clocks) while its SIMD partner issues
    // MFMAs back to back, and ~4 clocks inside the MFMA stream itself (tools/mfma_lds_probe, profiles/r06_mfma_lds_probe.txt): so LOAD
    // holds the fragment reads and nothing else -- their addresses are computed in the COMP phase before (addr_*), the DMA pieces
    // ride between the MFMAs.
    uint32_t addr_a[RA::NB], addr_b[RB_::NB];
#pragma unroll
    for (int j = 0; j < RA::NB; ++j) addr_a[j] = ra.base[j];
#pragma unroll
    for (int j = 0; j < RB_::NB; ++j) addr_b[j] = rb.base[j];
    FragU fa[NSLAB][MT], fb[NSLAB][NT];
    auto read_stage = [&]() {                      // every fragment of the stage addr_* points at, slab by slab
        if (no_reads) return;
        static_for<NSLAB>([&](auto sc) {
            constexpr int S = decltype(sc)::value;
            static_for<NRB>([&](auto rc) { rb.template emit_at<decltype(rc)::value, S>(fb[S], addr_b); });
            static_for<NRA>([&](auto rc) { ra.template emit_at<decltype(rc)::value, S>(fa[S], addr_a); });
        });
    };
    // COMP: the stage's MFMAs with (DMA ? this wave's pieces of its next stage : nothing) and the next stage's read addresses between them
    constexpr int NADDR = RA::NB + RB_::NB;
    auto comp_stage_ = [&](auto dmac, auto mfc, uint32_t slot, uint32_t nxt) {
        constexpr bool DMA = decltype(dmac)::value, MF = decltype(mfc)::value;
        constexpr int GB = G - GA;                  // pieces of this phase
        constexpr int NF = (DMA ? GB : 0) + NADDR;
        auto addr_add = [&](auto jc) {
            constexpr int J = decltype(jc)::value;
            if constexpr (J < RA::NB) pinned_add(addr_a[J], nxt, ra.base[J]);
            else pinned_add(addr_b[J - RA::NB], nxt, rb.base[J - RA::NB]);
        };
        static_for<NM>([&](auto mc) {
            constexpr int M = decltype(mc)::value, S = M / (MT * NT), Q = M % (MT * NT);
            if constexpr (MF) {
                if constexpr (S % NACC == 0) mma16_pinned(acc[Q / NT][Q % NT], fb[S][Q % NT].v, fa[S][Q / NT].v);
                else mma16_pinned(accx[S % NACC - 1][Q / NT][Q % NT], fb[S][Q % NT].v, fa[S][Q / NT].v);
            }
            if constexpr (SPREAD) {
                // the narrow tile has 16 MFMAs for the same six pieces: one behind (almost) every MFMA, each piece held the wave's issue for
                // ~30 clocks (COMP 464 clocks for 272 of MFMA, profiles/r06_pn_looptrace.txt) -- a piece every ~2.7 MFMAs is what the
                // 256 x 128 form has and pays nothing for
                constexpr int d0 = DMA ? (M * GB + NM - 1) / NM : 0, d1 = DMA ? ((M + 1) * GB + NM - 1) / NM : 0;
                static_for<d1 - d0>([&](auto fc) { dma_piece(std::integral_constant<int, GA + d0 + decltype(fc)::value>{}, slot); });
                constexpr int a0 = M * NADDR / NM, a1 = (M + 1) * NADDR / NM;
                static_for<a1 - a0>([&](auto fc) { addr_add(std::integral_constant<int, a0 + decltype(fc)::value>{}); });
            } else {
                constexpr int f0 = M * NF / NM, f1 = (M + 1) * NF / NM;
                static_for<f1 - f0>([&](auto fc) {
                    constexpr int F = f0 + decltype(fc)::value;
                    if constexpr (DMA && F < GB) dma_piece(std::integral_constant<int, GA + F>{}, slot);
                    else addr_add(std::integral_constant<int, F - (DMA ? GB : 0)>{});
                });
            }
        });
        if constexpr (DMA) dma_advance();
    };
    auto comp_stage = [&](auto dmac, uint32_t slot, uint32_t nxt) {
#ifdef MB_GEMM_ABLATE
        if (no_mfma) { comp_stage_(dmac, std::false_type{}, slot, nxt); return; }
#endif
        comp_stage_(dmac, std::true_type{}, slot, nxt);
    };
    typedef std::integral_constant<int, 1> W1;       // wait until only the newest stage is in flight
    typedef std::integral_constant<int, 0> W0;       // drain
    typedef std::integral_constant<int, -1> WN;      // nothing to wait for
    auto land = [&](auto wc_) {
        constexpr int W = decltype(wc_)::value;
        if constexpr (W == 1) wait_vmcnt<G>();
        else if constexpr (W == 0) wait_vmcnt<0>();
    };

    // Group 0 requests stage t+2 during COMP(t) (phase 2t+1), group 1 stage t+3 during ITS COMP(t) (phase 2t+2): the slot is the
    // one stage t lived in, which both groups have read by then.  Either way a wave's pieces have >= 2 phases to land.
    issue_stage(0u);
    issue_stage((uint32_t)STAGE);
    if (grp) { issue_half_b(2u * STAGE); wait_vmcnt<2 * G - GA>(); } else wait_vmcnt<G>();      // (group 1's COMP half runs a stage ahead)
    __builtin_amdgcn_s_barrier();                    // stage 0 has landed for everybody
    stamp(1);
    // ONE barrier per k-stage (MB_PP_ONE_BARRIER, default): group 0 meets it behind its COMP, group 1 behind its LOAD -- the pair that
    // carries the ring's hazards (a wave's share of stage t+1 has landed; the slot the next DMA pieces go to has been read by everybody).
    // The other pair of the two-barrier form (group 0 behind LOAD, group 1 behind COMP) only forced the alternation, which the first pair
    // restores every stage anyway: group 0 leaves it into a LOAD, group 1 into a COMP.  An eight-wave barrier costs ~200 clocks of a
    // ~750-clock phase (profiles/r06_pn_looptrace.txt).
    if (!ONEB && grp) __builtin_amdgcn_s_barrier();  // (two-barrier form) group 1 runs one phase behind
    uint32_t cur = 0u, dslot = grp ? 0u : 2u * STAGE, dslot_a = 2u * STAGE;
    // DM: 1 = every wave requests its next stage between the MFMAs, 2 = group 0 only (in front of the MFMAs), 0 = nobody
    auto stage = [&](int t, auto dm, auto wc_) {
        constexpr int DM = decltype(dm)::value;
        (void)t;
        // ---- LOAD(t)
        PP_LT(t, 0);
        read_stage();
        if constexpr (GA > 0 && DM != 0) issue_half_a(dslot_a);      // stage t+2, first half (both groups)
        PP_LT(t, 1);
        lds_wait_all();
        if constexpr (ONEB && GA > 0) { if (grp) land(wc_); }        // (group 0 has stage t+1 AND the half just issued in flight: no wait here)
        else land(wc_);                              // group 1: its share of stage t+1 has landed (group 0: nothing younger than stage t+1 yet)
        PP_LT(t, 2);
        __builtin_amdgcn_sched_barrier(0);           // the scalar bookkeeping of COMP stays behind the barrier (in LOAD it would cost ~16 clocks each)
        if (!ONEB || grp) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        PP_LT(t, 3);
        // ---- COMP(t)
        const uint32_t nxt = cur == 2u * STAGE ? 0u : cur + STAGE;
        if constexpr (DM == 2) { if (!grp) issue_half_b(dslot); }
        // the multiplying wave outranks its SIMD partner's reads: by age alone the older wave (group 0) wins BOTH ways and group 1's
        // 32 MFMAs take ~930 clocks instead of ~510 (profiles/r06_pp_looptrace.txt)
        if (!(p.dbg & 16)) __builtin_amdgcn_s_setprio(1);
        comp_stage(std::integral_constant<bool, DM == 1>{}, dslot, nxt);
        if (!(p.dbg & 16)) __builtin_amdgcn_s_setprio(0);
        PP_LT(t, 4);
        if constexpr (ONEB && GA > 0) { if (!grp) land(wc_); }
        else land(wc_);                              // group 0: its share of stage t+1 (group 1: of stage t+2, a phase early -- it has had two)
        PP_LT(t, 5);
        if (!ONEB || !grp) __builtin_amdgcn_s_barrier();
        cur = nxt;
        dslot = dslot == 2u * STAGE ? 0u : dslot + STAGE;
        dslot_a = dslot_a == 2u * STAGE ? 0u : dslot_a + STAGE;
    };
    typedef std::integral_constant<int, 0> D0;
    typedef std::integral_constant<int, 1> D1;
    typedef std::integral_constant<int, 2> D2;
    int t = 0;
    for (; t + 3 < nt; ++t) stage(t, D1{}, W1{});
    stage(t, D2{}, W1{});
    stage(t + 1, D0{}, W0{});
    stage(t + 2, D0{}, WN{});
    if (!ONEB && !grp) __builtin_amdgcn_s_barrier(); // (two-barrier form) pairs with group 1's last COMP
    if constexpr (NACC > 1) {
#pragma unroll
        for (int a = 0; a < NACC - 1; ++a)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] += accx[a][i][j];
    }
    stamp(2);
    if constexpr (KSW) gemm_epilogue<T, BM, BN, MODE, true, NW, 2>(p, acc, m0, n0, wave, lane, smem, pre);
    else gemm_epilogue<T, BM, BN, MODE, false, NW>(p, acc, m0, n0, wave, lane, smem, pre);
    if (p.trace) {
        stamp(3);
        wait_vmcnt<0>();
        stamp(4);
#ifdef MB_GEMM_LOOPTRACE
        if ((tid & 255) == 0) {
            unsigned long long* dst = p.trace + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * kTraceStride + 8 + grp * kPpIters * kPpPoints;
            for (int i = 0; i < kPpIters * kPpPoints; ++i) dst[i] = lt[grp][i];
        }
#endif
    }
#undef PP_LT
}

template <bool AK, bool BK, int MODE>
__global__ void __launch_bounds__(512) gemm_pp_kernel(const GemmArgs p) {
    __shared__ __attribute__((aligned(1024))) char smem[kPpSlots * kPpStage];
    int m0, n0;
    if (!tile_origin<kPpBM, kPpBN>(p, m0, n0, blockIdx.x)) return;
    gemm_pp_body<kPpBM, kPpBN, kPpKB, AK, BK, MODE>(p, m0, n0, smem);
}

template <bool AK, bool BK, int MODE>
__global__ void __launch_bounds__(512) gemm_pn_kernel(const GemmArgs p) {
    __shared__ __attribute__((aligned(1024))) char smem[kPpSlots * kPpStage];
    int m0, n0;
    if (!tile_origin<kPnBM, kPnBN>(p, m0, n0, blockIdx.x)) return;
    gemm_pp_body<kPnBM, kPnBN, kPnKB, AK, BK, MODE, kPnKsw>(p, m0, n0, smem);
}

// a narrow dgrad launch (dX = dY W + R) with riders: 228 tiles at T = 2400 leave whole CUs idle (16 once the grid's padding is counted
// per XCD); the riders come FIRST in the grid (multiples of 8: the tiles keep their XCDs) and each takes a CU to itself
__global__ void __launch_bounds__(512) gemm_pn_ride_kernel(const GemmArgs p, const AdamRide ride) {
    __shared__ __attribute__((aligned(1024))) char smem[kPpSlots * kPpStage];
    if ((int)blockIdx.x < ride.blocks) {
        adam_ride_block<512, MB_RIDE_UNR>(ride, (int)blockIdx.x);
        return;
    }
    int m0, n0;
    if (!tile_origin<kPnBM, kPnBN>(p, m0, n0, (int)blockIdx.x - ride.blocks)) return;
    gemm_pp_body<kPnBM, kPnBN, kPnKB, false, true, EPI_ADD_RES, kPnKsw>(p, m0, n0, smem);
}

template <bool AK, bool BK, int MODE>
__global__ void __launch_bounds__(512) gemm_pt_kernel(const GemmArgs p) {
    __shared__ __attribute__((aligned(1024))) char smem[kPpSlots * (kPtBM + kPtBN) * kPtKB];
    int m0, n0;
    if (!tile_origin<kPtBM, kPtBN>(p, m0, n0, blockIdx.x)) return;
    gemm_pp_body<kPtBM, kPtBN, kPtKB, AK, BK, MODE>(p, m0, n0, smem);
}
__global__ void __launch_bounds__(512) gemm_pt_ride_kernel(const GemmArgs p, const AdamRide ride) {
    __shared__ __attribute__((aligned(1024))) char smem[kPpSlots * (kPtBM + kPtBN) * kPtKB];
    if ((int)blockIdx.x < ride.blocks) {
        adam_ride_block<512, MB_RIDE_UNR>(ride, (int)blockIdx.x);
        return;
    }
    int m0, n0;
    if (!tile_origin<kPtBM, kPtBN>(p, m0, n0, (int)blockIdx.x - ride.blocks)) return;
    gemm_pp_body<kPtBM, kPtBN, kPtKB, false, true, EPI_ADD_RES>(p, m0, n0, smem);
}

// the weight gradients of a layer (dW = dY^T X: both operands k-major), one launch (gemm.hip: launch_grouped places the tiles)
__global__ void __launch_bounds__(512) gemm_pp_grouped_tn_kernel(const GroupedGemmArgs ga) {
    __shared__ __attribute__((aligned(1024))) char smem[kPpSlots * kPpStage];
    if ((int)blockIdx.x < ga.ride.blocks) {          // rider: an optimizer update on a CU that has no tile (kernels.h AdamRide)
        adam_ride_block<512, MB_RIDE_UNR>(ga.ride, (int)blockIdx.x);
        return;
    }
    int g, m0, n0;
    if (!grouped_tile_origin<kPpBM, kPpBN>(ga, g, m0, n0)) return;
    gemm_pp_body<kPpBM, kPpBN, kPpKB, true, true, EPI_ACCUM_F32>(ga.g[g], m0, n0, smem);
}

int gemm_pp_launch(bool ak, bool bk, int mode, const GemmArgs& p, dim3 grid, hipStream_t st) {
#define MB_PP(AKV, BKV, MODEV) \
    if (ak == AKV && bk == BKV && mode == MODEV) { \
        MB_GEMM_LAUNCH((gemm_pp_kernel<AKV, BKV, MODEV>), grid, dim3(512), st, p, &p, 1); \
        return (int)hipGetLastError(); \
    }
    MB_PP(false, false, EPI_BIAS)
    MB_PP(false, false, EPI_BIAS_GELU)
    MB_PP(false, true, EPI_DGELU)
    MB_PP(true, true, EPI_ACCUM_F32)
#undef MB_PP
    return MB_ERR_MODE;
}

// the 128 x 64 form (bf16; K a multiple of 128, at least three stages): the N = 768 forward and dgrad launches
int gemm_pn_launch(bool ak, bool bk, int mode, const GemmArgs& p, dim3 grid, hipStream_t st, bool tall) {
#define MB_PN(AKV, BKV, MODEV) \
    if (ak == AKV && bk == BKV && mode == MODEV) { \
        if (tall) MB_GEMM_LAUNCH((gemm_pt_kernel<AKV, BKV, MODEV>), grid, dim3(512), st, p, &p, 1); \
        else MB_GEMM_LAUNCH((gemm_pn_kernel<AKV, BKV, MODEV>), grid, dim3(512), st, p, &p, 1); \
        return (int)hipGetLastError(); \
    }
    MB_PN(false, false, EPI_BIAS)
    MB_PN(false, false, EPI_BIAS_DROP_RES)
    MB_PN(false, false, EPI_ADD_RES)
    MB_PN(false, true, EPI_ADD_RES)
#undef MB_PN
    return MB_ERR_MODE;
}

int gemm_pn_ride_launch(const GemmArgs& p, const AdamRide& ride, dim3 grid, hipStream_t st, bool tall) {
    gemm_log_ride(ride);
    if (tall) {
        gemm_log((const void*)gemm_pt_ride_kernel, st, &p, 1);
        hipLaunchKernelGGL(gemm_pt_ride_kernel, dim3(grid.x + ride.blocks), dim3(512), 0, st, p, ride);
    } else {
        gemm_log((const void*)gemm_pn_ride_kernel, st, &p, 1);
        hipLaunchKernelGGL(gemm_pn_ride_kernel, dim3(grid.x + ride.blocks), dim3(512), 0, st, p, ride);
    }
    return (int)hipGetLastError();
}

int gemm_pp_grouped_launch(const GroupedGemmArgs& ga, int grid, hipStream_t st) {
    MB_GEMM_LAUNCH(gemm_pp_grouped_tn_kernel, dim3(grid + ga.ride.blocks), dim3(512), st, ga, ga.g, ga.count);
    gemm_log_ride(ga.ride);
    return (int)hipGetLastError();
}

}  // namespace mb
