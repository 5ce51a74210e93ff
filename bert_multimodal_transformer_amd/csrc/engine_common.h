// Host-side helpers shared by the MAG-BERT and MAG-XLNet step executors (engine.hip, xlnet_engine.hip).
#pragma once
#include <string>
#include <vector>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <algorithm>
#include "kernels.h"
#include "../../include/magbert_hip.h"

using namespace mb;

namespace {

struct TensorInfo {
    std::string name;
    size_t off, numel;
    int ndim;
    int64_t shape[4];
    int decay;
};

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

inline DropKey make_key(uint64_t seed, uint64_t step, uint32_t site, float p) {
    DropKey k;
    k.dyn = nullptr;
    if (!(p > 0.f)) { k.k0 = k.k1 = k.thresh = 0; k.scale = 1.f; return k; }
    const uint64_t h = splitmix64(splitmix64(seed) ^ splitmix64(step * 0x100000001B3ull + site));
    k.k0 = (uint32_t)h;
    k.k1 = (uint32_t)(h >> 32);
    double t = (double)p * 4294967296.0;
    if (t > 4294967295.0) t = 4294967295.0;
    k.thresh = (uint32_t)(t + 0.5);
    k.scale = 1.0f / (1.0f - p);
    return k;
}
const DropKey kNoDrop = {0u, 0u, 0u, 1.0f, nullptr};
inline DropKey dk(const mb_dropkey* d) {
    if (!d) return kNoDrop;
    DropKey k = {d->k0, d->k1, d->thresh, d->scale, nullptr};
    return k;
}

enum { SITE_EMB = 0, SITE_MAG = 1, SITE_HEAD = 2, SITE_LAYER0 = 16 };   // layer l: 16+4l+{0 attn probs,1 attn out,2 ffn out}

inline size_t esize(int dtype) { return dtype == DT_BF16 ? 2 : 4; }

struct Carver {
    size_t off = 0;
    size_t take(size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; }
};

// workspace of the MAG operator: saved activations + scratch (shared by the op-level API and the engine)
inline int mag_wgrad_tile() {
    static int t = -1;
    if (t < 0) { const char* v = getenv("MB_MAG_WGRAD_TILE"); t = (v && atoi(v) == 128) ? 128 : 64; }
    return t;
}

// ring depth of MAG's grouped weight-gradient launch (64 x 64 tiles only): MB_MAG_WGRAD_STAGES = 2 | 3 | 4 | 5
inline int mag_wgrad_stages() {        // (read per call -- capture time only -- so that a test can compare the variants in one process)
    const char* v = getenv("MB_MAG_WGRAD_STAGES");
    return v ? atoi(v) : 4;
}

struct MagWs {
    int Vp, Ap;
    size_t We, Wv, Wa, vp, ap, Ze, Zv, Za, mean, rstd;                       // forward (saved)
    size_t dZe, dZv, dZa, dep, dWe, dWv, dWa, dvp, dap;                      // backward scratch
    size_t bytes;
    void init(int dtype, int T_, int H, int V, int A) {
        const size_t es = esize(dtype);
        const size_t T = align_up((size_t)T_, 64);   // token rows padded to the GEMM k-tile (pad rows stay zero)
        // modality widths padded to the tile of the grouped weight-gradient launch: 64 -- three problems of 288 + 24 + 48 tiles whose
        // K = T loop of 38 stages is pure latency (35 us for 7 GFLOP).  MB_MAG_WGRAD_TILE=128 (72 + 12 + 12 tiles of 19 stages, one
        // per CU) was measured SLOWER: ~48 us in the kernel trace, +12 .. 28 us per step (profiles/r03_oneoffs_ab.txt)
        const int tile = mag_wgrad_tile();
        Vp = (V + tile - 1) / tile * tile;
        Ap = (A + tile - 1) / tile * tile;
        Carver c;
        We = c.take((size_t)2 * H * H * es); Wv = c.take((size_t)2 * H * Vp * es); Wa = c.take((size_t)2 * H * Ap * es);
        vp = c.take((size_t)T * Vp * es); ap = c.take((size_t)T * Ap * es);
        Ze = c.take((size_t)T * 2 * H * es); Zv = c.take((size_t)T * 2 * H * es); Za = c.take((size_t)T * 2 * H * es);
        mean = c.take((size_t)T * 4); rstd = c.take((size_t)T * 4);
        dZe = c.take((size_t)T * 2 * H * es); dZv = c.take((size_t)T * 2 * H * es); dZa = c.take((size_t)T * 2 * H * es);
        dep = c.take((size_t)T * H * es);
        dWe = c.take((size_t)2 * H * H * 4); dWv = c.take((size_t)2 * H * Vp * 4); dWa = c.take((size_t)2 * H * Ap * 4);
        dvp = c.take((size_t)T * Vp * 4); dap = c.take((size_t)T * Ap * 4);
        bytes = c.off;
    }
};

#define CK(x) do { int _e = (x); if (_e) { mb::ck_trace(#x, __FILE__, __LINE__, _e); return _e; } } while (0)

inline int gemm(int dtype, int layout, int mode, int M, int N, int K, const void* A, int lda, const void* B, int ldb, void* C,
         int ldc, void* C2, float* Cf, const float* bias, const void* R, int ldr, DropKey drop, int splits, int tile,
         hipStream_t st, int bseg = 0, size_t bseg_stride = 0, GradAcc acc = {}) {
    GemmArgs a = {};
    a.acc = acc;
    a.bseg = bseg; a.bseg_stride = bseg_stride;        // segmented B (kernels.h): pieces of B's contiguous dimension in separate tensors
    a.A = A; a.B = B; a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb;
    a.C = C; a.ldc = ldc; a.C2 = C2; a.Cf = Cf; a.bias = bias; a.R = R; a.ldr = ldr; a.alpha = 1.0f; a.drop = drop;
    a.kchunk = K; a.colsum = nullptr;
    if (mode == EPI_DGELU) { a.colsum = Cf; a.Cf = nullptr; }     // EPI_DGELU: the fp32 pointer carries the fused bias grad
    return gemm_launch(dtype, layout, mode, a, splits, tile, st);
}

// dX = dY . W + R (EPI_ADD_RES) or (dY . W) * gelu'(R) with the bias gradient `colsum` (EPI_DGELU), GEMM_NN: the plain launch, or -- when
// riders are active, the launch is one of the two kernels that leave block slots free (kernels.h gemm_nn_ride_tiles) and `take(budget,
// blocks)` hands out a piece of the optimizer update (kernels.h AdamRide) -- the same launch with rider workgroups behind its tiles.
// Budgets: what the free slots stream while the tiles multiply, by the launch's size: 1.25 M parameters next to 11.3 GFLOP for the 64 x 64
// kernel (the riders share SIMDs with the tiles: every instruction of theirs costs the partner wave an MFMA slot, so the gain is small and
// turns at ~1.5 M: same box 3.566 ms without | 3.544 at 1 - 1.5 M | 3.578 at 2.5 M per launch), 14,336 per free slot of the 128 x 128
// kernel (56 CUs that hold one tile instead of two)  (profiles/r06_adamw_ride_dgrad.txt)
inline long ride_pn_params() {       // MB_ADAMW_RIDE_PN_PARAMS: parameters per narrow dgrad launch on the 128 x 64 tile (0 = by the launch's size)
    static long v = -1;
    if (v < 0) { const char* e = getenv("MB_ADAMW_RIDE_PN_PARAMS"); v = e ? atol(e) : 0; }
    return v;
}
struct RideOpts { int dgrad = 2, dgrad_blocks = 0; long dgrad_params = 0, dgelu_params = 0; };
template <class Take>
inline int dgrad_with_riders(int dt, int mode, int Mo, int No, int Ko, const void* dY, int ldy, const void* Wt, int ldw, void* dX, int ldx, const void* R,
                             int ldr, float* colsum, DropKey drop, GradAcc acc, hipStream_t st, bool active, const RideOpts& ro, int cus, Take&& take) {
    if (active && ro.dgrad && (mode == EPI_ADD_RES || ro.dgrad >= 2)) {
        GemmArgs a = {};
        a.A = dY; a.B = Wt; a.M = Mo; a.N = No; a.K = Ko; a.lda = ldy; a.ldb = ldw; a.C = dX; a.ldc = ldx; a.R = R; a.ldr = ldr;
        a.alpha = 1.0f; a.drop = drop; a.kchunk = Ko; a.colsum = colsum; a.acc = acc;
        int per_cu = 0;
        const int tiles = gemm_nn_ride_tiles(dt, mode, a, &per_cu);
        const int slots = per_cu * cus, rounds = tiles > 0 ? (tiles + slots - 1) / slots : 0;      // (the riders take what the LAST round leaves free)
        const int blocks = tiles > 0 ? (ro.dgrad_blocks > 0 ? ro.dgrad_blocks : rounds * slots - tiles) / 8 * 8 : 0;
        if (blocks >= 8) {
            size_t budget = ro.dgrad_params > 0 ? (size_t)ro.dgrad_params : (size_t)(1.25e6 * ((double)Mo * No * Ko) / (2400.0 * 768.0 * 3072.0)) / 1024 * 1024;
            if (mode == EPI_DGELU) budget = ro.dgelu_params > 0 ? (size_t)ro.dgelu_params : (size_t)blocks * 14336;
            // the 128 x 64 ping-pong tile (one per CU): the riders are whole idle CUs, ~41 GB/s each for the launch's duration
            else if (per_cu == 1) budget = ride_pn_params() > 0 ? (size_t)ride_pn_params() : (size_t)(blocks * 16384.0 * ((double)Mo * No * Ko) / (2400.0 * 768.0 * 3072.0)) / 1024 * 1024;
            // (16 K parameters per rider CU next to 11.3 GFLOP: more stretches the launch -- 26 K: +1 % per step, 40 K: +6 %; none at all: +0.4 %;
            //  profiles/r06_pn_default_ab.txt)
            const AdamRide r = take(budget, blocks);
            if (r.blocks) return gemm_nn_ride_launch(dt, mode, a, r, st);
        }
    }
    if (mode == EPI_DGELU)
        return gemm(dt, GEMM_NN, EPI_DGELU, Mo, No, Ko, dY, ldy, Wt, ldw, dX, ldx, nullptr, colsum, nullptr, R, ldr, drop, 1, 0, st, 0, 0, acc);
    return gemm(dt, GEMM_NN, EPI_ADD_RES, Mo, No, Ko, dY, ldy, Wt, ldw, dX, ldx, nullptr, nullptr, nullptr, R, ldr, drop, 1, 0, st);
}

// operands of one weight gradient dW[Mo][No] += dY[rows][Mo]^T X[rows][No] (GEMM_TN, EPI_ACCUM_F32)
inline GemmArgs wgrad_args(int Mo, int No, int rows, const void* dY, int ldy, const void* X, int ldx, float* dW, int ldw) {
    GemmArgs a = {};
    a.A = dY; a.B = X; a.M = Mo; a.N = No; a.K = rows; a.lda = ldy; a.ldb = ldx;
    a.C = nullptr; a.ldc = ldw; a.C2 = nullptr; a.Cf = dW; a.bias = nullptr; a.colsum = nullptr; a.R = nullptr; a.ldr = 0;
    a.alpha = 1.0f; a.drop = kNoDrop; a.kchunk = rows; a.dbg = 0; a.reg_m = a.reg_n = a.tpr_m = a.tpr_n = 0;
    return a;
}

// wgrad: dW[N'][K'] += dY^T X, reduction over `rows` tokens; picks tile + split-K to fill the 256 CUs
inline int wgrad(int dtype, int Mo, int No, int rows, const void* dY, int ldy, const void* X, int ldx, float* dW, int ldw,
          hipStream_t st) {
    const long t128 = (long)((Mo + 127) / 128) * ((No + 127) / 128);
    const long t64 = (long)((Mo + 63) / 64) * ((No + 63) / 64);
    // split-K costs one fp32 atomic per output element per split (measured: 4.7 M atomics ~ 50 us), so it is used only
    // when the tile grid alone cannot occupy the chip
    int tile, splits;
    if (t128 >= 256) { tile = 128; splits = 1; }
    else if (t64 >= 100) { tile = 64; splits = 1; }        // measured: [768x3072] K=2432: 64^2 32 us vs 128^2 52 us
    else { tile = 64; splits = (int)((256 + t64 - 1) / t64); }
    if (splits > 8) splits = 8;
    while (splits > 1 && rows / splits < 128) --splits;
    if (splits < 1) splits = 1;
    (void)t128;
    return gemm(dtype, GEMM_TN, EPI_ACCUM_F32, Mo, No, rows, dY, ldy, X, ldx, nullptr, ldw, nullptr, dW, nullptr, nullptr, 0,
                kNoDrop, splits, tile, st);
}

// ------------------------------------------------------------------------------------------------ whole-step machinery
// Shared by both engines (mb_bert_train_step / mb_xlnet_train_step, mb_*_load_batch): everything that changes from one optimizer
// step to the next lives in the workspace -- staging buffers for the six batch tensors, the dropout keys of every site, the AdamW
// scalars of the two parameter groups -- written by ONE prologue launch; the rest of the step is a fixed kernel sequence that can
// be replayed as a hipGraph.
struct StepGraph {
    int B, L, with_opt, overwrite; const void *logits, *loss, *loss_run, *m, *v; float loss_scale; hipStream_t st;
    int nseg;                                       // the step as `nseg` linear graphs launched back to back (1 = the whole step)
    int variant;                                    // which segmentation (0 = the single-process step, else a hash of the data-parallel plan)
    const void* tag;                                // whose events the graphs' nodes record (the mb_comm of a data-parallel step), else null
    std::vector<hipGraph_t> graph; std::vector<hipGraphExec_t> exec;
    void destroy() {
        for (auto x : exec) if (x) hipGraphExecDestroy(x);
        for (auto g : graph) if (g) hipGraphDestroy(g);
        exec.clear(); graph.clear();
    }
};

// one captured pass of a stage-driven (data-parallel) step: stage -1 = forward, 0 .. = backward stage
struct StageGraph { int B, L, stage, overwrite; const void *logits, *loss, *loss_run; float loss_scale; hipStream_t st; hipGraph_t graph; hipGraphExec_t exec; };

struct StepMixin {
    bool dyn = false;              // dropout keys / AdamW scalars are read from device memory (set while a train step is built)
    bool stage_mode = false;       // a stage-driven step is being built: every stage's gradients must be final when the stage returns
    std::vector<StageGraph> stage_graphs;
    // single-call step: the step prologue converts the two modality tensors straight into MAG's packed GEMM operands (set by the
    // engines: workspace offsets of the operands); `packed` tells mag_fwd_impl of the step that the pack_pad launches already happened
    bool pk_enable = false, packed = false;
    // ... and MAG's weight operands too (PrologueArgs::MagPackW, extra blocks of the same launch; the engines fill `pkw` when buffers are
    // bound): the first kernel of every replayed step used to be that pack.  MB_PROLOGUE_PACKW=0: back in the graph
    bool pkw_enable = false, packed_w = false;
    PrologueArgs::MagPackW pkw = {};
    size_t pk_vis = 0, pk_aco = 0; int pk_Vp = 0, pk_Ap = 0, pk_dtype = 0;
    // single-call step: the prologue counts the occurrences of every token id (workspace offset of the table, 0 = none); `counted`
    // tells the engine's backward that the table describes the batch of this step
    size_t idcnt_off = 0; bool idcnt_enable = false, counted = false;
    bool loss_cleared = false;     // single-call step: the step prologue clears the loss accumulator (no zero_fill launch in the forward)
    bool capturing = false;        // the stream is in capture mode: nothing outside the captured sequence may be waited for
    int nsites = 0;
    size_t ws_state = 0, ws_in_ids = 0, ws_in_seg = 0, ws_in_mask = 0, ws_in_vis = 0, ws_in_aco = 0, ws_in_lab = 0;
    std::vector<StepGraph> graphs;
    size_t graph_launches = 0, graph_captures = 0;
    // Captures run on a stream of the engine's own, never on the caller's: hipEventQuery on an event whose last record was on a
    // stream that is capturing NOW fails and invalidates that capture (ROCm 7.2: hipErrorCapturedEvent), and a host framework may
    // poll such events from another thread at any time -- PyTorch's NCCL watchdog does, for collectives it ran on the caller's
    // stream (found by the one-rank RCCL test: a race between its 100 ms poll and the first capture).  A graph is launch-stream
    // agnostic, so the captured sequence is replayed on the caller's stream as before.
    hipStream_t cap = nullptr;
    int capture_stream(hipStream_t* out) {
        if (!cap) CK((int)hipStreamCreateWithFlags(&cap, hipStreamNonBlocking));
        *out = cap;
        return MB_OK;
    }
    // Known-zero gradients.  The fused AdamW leaves the flat gradient buffer zeroed (optimizer.zero_grad()); when nothing has
    // written to it since, the layer weight gradients of the next backward are STORED instead of accumulated: the read half of
    // a 340 MB read-modify-write per step (28 MB per layer, straight from HBM: AdamW streamed the zeros out non-temporally).
    // grads_zero: set by a step that ran the optimizer and by mb_*_mark_grads_zero (the host zeroed the buffer itself);
    // consumed -- and cleared -- by the first stage of the next backward.  A false "known zero" is the only way this can go
    // wrong, so everything that is not the engine's own zeroing leaves it false.  MB_WGRAD_OVERWRITE=0 turns it off.
    bool grads_zero = false, ow_pass = false, in_step = false;
    int ow_enable = 1;
    // Lazy zeroing (round 3).  The zeros the fused AdamW writes over the layers' GEMM weight gradients (340 MB per step) are only
    // ever overwritten by the next backward, so a single-call step that ends with the optimizer leaves that range as it is:
    // "logically zero, physically stale" (grads_stale, [stale_begin, stale_end)).  Whoever else is about to look at it gets real
    // zeros first: a backward that accumulates (begin_backward_pass / train_step_impl), mb_*_materialize_grads (called by the
    // Python mirror before optimizer.step(), flat_grads, mark_grads_zero(False)).  MB_ADAMW_KEEP=0 turns it off.
    // deterministic mode (MB_DETERMINISTIC=1, common.h GradAcc): multi-writer gradient sums go to a 64-bit fixed-point shadow of
    // G[det_begin, n) in the workspace and are folded into the fp32 gradients at the end of the backward
    int deterministic = 0;
    size_t ws_det = 0, det_begin = 0, det_end = 0;
    GradAcc acc_of(char* ws, const float* G) const {
        GradAcc a = {nullptr, nullptr};
        if (deterministic && ws && G) { a.shadow = (long long*)(ws + ws_det) - det_begin; a.base = G; }
        return a;
    }
    int cus = 0;                                       // compute units of the current device (AdamRide: slots a grouped launch leaves empty)
    int cu_count() {
        if (cus <= 0) {
            int dev = 0, n = 0;
            if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess) cus = n;
            if (cus <= 0) cus = 256;
        }
        return cus;
    }
    bool grads_stale = false, ow_covers = false;       // ow_covers: the overwriting (grouped) launch covers [stale_begin, stale_end)
    size_t stale_begin = 0, stale_end = 0;
    int keep_enable = 1;
    bool keep_in_step() const { return keep_enable && ow_enable && ow_covers && stale_end > stale_begin; }
    int materialize_grads(float* G, hipStream_t st) {
        if (grads_stale && G) CK(zero_fill(G + stale_begin, (stale_end - stale_begin) * sizeof(float), st));
        grads_stale = false;
        return MB_OK;
    }
    int begin_backward_pass(float* G, hipStream_t st) {
        if (in_step) return MB_OK;           // the single-call step decided already (and replays a graph captured for that decision)
        ow_pass = grads_zero && ow_enable;
        grads_zero = false;
        if (!(ow_pass && ow_covers)) CK(materialize_grads(G, st));      // this pass adds onto the range: it needs the zeros
        grads_stale = false;                 // (else: every layer stage of the pass stores its gradients over the stale values)
        return MB_OK;
    }

    // measurement hooks (bench.py): timing events around every grouped weight-gradient launch and around the optimizer launches
    // of a single-call step; steps run launch by launch while they are on (events cannot live inside a captured graph)
    bool prof = false;
    std::vector<hipEvent_t> pev;   // [2 * layers] grouped weight gradients, then [2] AdamW
    int prof_layers = 0;
    int set_profiling(int on, int layers) {
        if (on && pev.empty()) {
            prof_layers = layers;
            pev.assign((size_t)2 * layers + 2, nullptr);
            for (auto& ev : pev) CK((int)hipEventCreate(&ev));
        }
        prof = on != 0;
        return MB_OK;
    }
    int prof_mark(int idx, hipStream_t st) { return (prof && !capturing && idx < (int)pev.size()) ? (int)hipEventRecord(pev[idx], st) : 0; }
    int prof_span_us(int first, int pairs, float* avg_us) {
        if (!prof || !avg_us || pev.empty()) return MB_ERR_ARG;
        double sum = 0.0;
        for (int i = 0; i < pairs; ++i) {
            float ms = 0.f;
            CK((int)hipEventSynchronize(pev[first + 2 * i + 1]));
            CK((int)hipEventElapsedTime(&ms, pev[first + 2 * i], pev[first + 2 * i + 1]));
            sum += ms;
        }
        *avg_us = (float)(sum * 1e3 / pairs);
        return MB_OK;
    }
    void destroy_prof() { for (auto& ev : pev) if (ev) hipEventDestroy(ev); pev.clear(); }

    void carve_step(Carver& w, size_t Tpad, int V, int A, int max_batch, int num_labels, int nsites_) {
        nsites = nsites_;
        ws_state = w.take(2 * sizeof(AdamArgs) + (size_t)nsites * 8);
        ws_in_ids = w.take(Tpad * 8); ws_in_seg = w.take(Tpad * 8); ws_in_mask = w.take(Tpad * 8);
        ws_in_vis = w.take(Tpad * (size_t)V * 4); ws_in_aco = w.take(Tpad * (size_t)A * 4);
        ws_in_lab = w.take((size_t)max_batch * num_labels * 4);
    }
    void drop_graphs() {
        if (cap) { hipStreamDestroy(cap); cap = nullptr; }
        for (auto& g : graphs) g.destroy();
        graphs.clear();
        for (auto& g : stage_graphs) { hipGraphExecDestroy(g.exec); hipGraphDestroy(g.graph); }
        stage_graphs.clear();
    }
    // replay (capturing on first use) one pass of a stage-driven step; enqueue(st) issues its kernels with dyn keys
    template <class Enq>
    int run_stage_graph(int B, int L, int stage, int overwrite, const void* logits, const void* loss, const void* loss_run, float loss_scale,
                        int mode, hipStream_t st, Enq enqueue) {
        struct Scope { StepMixin* m; ~Scope() { m->dyn = false; m->capturing = false; m->in_step = false; m->stage_mode = false; } } scope{this};
        dyn = true; in_step = true; stage_mode = true;
        if (mode == 2) return enqueue(st);
        StageGraph* g = nullptr;
        for (auto& x : stage_graphs)
            if (x.B == B && x.L == L && x.stage == stage && x.overwrite == overwrite && x.logits == logits && x.loss == loss && x.loss_run == loss_run &&
                x.loss_scale == loss_scale && x.st == st) { g = &x; break; }
        if (!g) {
            if (stage_graphs.size() >= 256) {
                hipGraphExecDestroy(stage_graphs.front().exec); hipGraphDestroy(stage_graphs.front().graph);
                stage_graphs.erase(stage_graphs.begin());
            }
            StageGraph ng = {B, L, stage, overwrite, logits, loss, loss_run, loss_scale, st, nullptr, nullptr};
            hipStream_t cs = nullptr;
            CK(capture_stream(&cs));
            CK((int)hipStreamBeginCapture(cs, hipStreamCaptureModeRelaxed));
            capturing = true;
            const int r = enqueue(cs);
            capturing = false;
            const int r2 = (int)hipStreamEndCapture(cs, &ng.graph);
            if (r) { if (ng.graph) hipGraphDestroy(ng.graph); return r; }
            CK(r2);
            CK((int)hipGraphInstantiate(&ng.exec, ng.graph, nullptr, nullptr, 0));
            stage_graphs.push_back(ng);
            g = &stage_graphs.back();
            ++graph_captures;
        }
        CK((int)hipGraphLaunch(g->exec, st));
        ++graph_launches;
        return MB_OK;
    }
    AdamArgs* adam_state(char* ws) const { return (AdamArgs*)(ws + ws_state); }
    uint32_t* key_state(char* ws) const { return (uint32_t*)(ws + ws_state + 2 * sizeof(AdamArgs)); }
    DropKey step_key(char* ws, bool training, uint64_t seed, uint64_t step, uint32_t site, float p) const {
        if (!training) return kNoDrop;
        if (!dyn) return make_key(seed, step, site, p);
        DropKey k = make_key(0, 0, site, p);          // thresh / scale of this site; (k0, k1) come from the device table
        if (k.thresh) { k.k0 = k.k1 = 0u; k.dyn = key_state(ws) + 2 * (size_t)site; }
        return k;
    }
    void fill_copies(PrologueArgs& pa, char* ws, const void* ids, const void* vis, const void* aco, const void* mask, const void* seg,
                     const void* lab, int B, int L, int V, int A, int num_labels, bool pack_modalities = false) const {
        const size_t T = (size_t)B * L;
        auto cp = [&](const void* src, size_t off, size_t bytes) {
            pa.src[pa.ncopies] = (const uint32_t*)src; pa.dst[pa.ncopies] = (uint32_t*)(ws + off); pa.dwords[pa.ncopies] = (uint32_t)(bytes / 4);
            ++pa.ncopies;
        };
        cp(ids, ws_in_ids, T * 8); cp(seg, ws_in_seg, T * 8); cp(mask, ws_in_mask, T * 8);
        if (pack_modalities) {
            pa.pack[0] = {(const float*)vis, ws + pk_vis, (int)T, V, pk_Vp, pk_dtype};
            pa.pack[1] = {(const float*)aco, ws + pk_aco, (int)T, A, pk_Ap, pk_dtype};
            pa.npack = 2;
        } else {
            cp(vis, ws_in_vis, T * V * 4); cp(aco, ws_in_aco, T * A * 4);
        }
        (void)num_labels;          // one float per sample either way: the regression target (num_labels == 1) or the class index
        if (lab) cp(lab, ws_in_lab, (size_t)B * 4);
    }
    void staged(char* ws, bool with_labels, const void** six) const {
        six[0] = ws + ws_in_ids; six[1] = ws + ws_in_vis; six[2] = ws + ws_in_aco; six[3] = ws + ws_in_mask; six[4] = ws + ws_in_seg;
        six[5] = with_labels ? ws + ws_in_lab : nullptr;
    }
};

// enqueue(logits, loss, loss_run, m, v, loss_scale, st): forward (reading the staged batch) + backward + AdamW of one engine
struct NoBetween { int operator()(int, hipStream_t) const { return MB_OK; } };
template <class E, class Enqueue, class Between = NoBetween>
inline int train_step_impl(E* e, char* ws, int V, int A, int num_labels, const void* ids, const void* vis, const void* aco, const void* mask,
                           const void* seg, const void* labels, int B, int L, uint64_t seed, uint64_t step, float* logits, float* loss,
                           float* loss_run, float* m, float* v, float lr, float beta1, float beta2, float eps, float weight_decay,
                           int opt_step, int correct_bias, float grad_scale, float loss_scale, int mode, bool force_launches,
                           hipStream_t st, Enqueue enqueue_inner, int nseg = 1, Between between = Between(), int variant = 0,
                           const void* tag = nullptr, int (*verify)(const void* tag, int nseg, int sg, hipGraph_t gr) = nullptr) {
    // The step may be cut into `nseg` segments: enqueue_inner(seg, ...) issues the kernels of one (captured and replayed as its own
    // LINEAR graph), between(seg, st) runs on the host right after segment `seg` was enqueued and is never captured -- the place for
    // cross-stream events (a graph with a fork inside runs on ROCm 7.2's slow path; a chain of linear graphs does not).
    // what this step's backward may assume about the gradient buffer -- part of the graph's identity -- and what it leaves behind
    const int ow = (e->grads_zero && e->ow_enable) ? 1 : 0;
    if (!(ow && e->ow_covers)) CK(e->materialize_grads(e->G, st));      // an accumulating backward needs the zeros that were skipped
    const bool stale_before = e->grads_stale;
    // what the step leaves behind is only known once everything was enqueued: an early error return leaves "nothing known"
    struct Flags {
        E* e; bool ok, with_opt, stale_before;
        ~Flags() {
            e->in_step = false; e->loss_cleared = false; e->packed = false; e->packed_w = false; e->counted = false;
            e->grads_zero = ok && with_opt;
            e->grads_stale = ok ? (with_opt && e->keep_in_step()) : stale_before;
        }
    } flags{e, false, m != nullptr, stale_before};
    e->in_step = true; e->ow_pass = ow != 0;
    auto enqueue = [&](int sg, float* lg, float* ls, float* lr_, float* m_, float* v_, float sc, hipStream_t s) {
        return enqueue_inner(sg, lg, ls, lr_, m_, v_, sc, s);
    };
    PrologueArgs pa = {};
    e->fill_copies(pa, ws, ids, vis, aco, mask, seg, labels, B, L, V, A, num_labels, e->pk_enable);
    e->packed = e->pk_enable;
    if (e->pkw_enable && e->pkw.W_hv) pa.magw = e->pkw;
    e->packed_w = pa.magw.W_hv != nullptr;
    pa.seed = seed; pa.step = step; pa.keys = e->key_state(ws); pa.nsites = e->nsites;
    pa.zero_dw = (uint32_t*)loss; e->loss_cleared = loss != nullptr;
    if (e->idcnt_enable && ids) { pa.ids = (const int64_t*)ids; pa.n_ids = B * L; pa.id_count = (int*)(ws + e->idcnt_off); }
    e->counted = pa.id_count != nullptr;
    if (m) {
        double ss = lr;
        if (correct_bias) ss = (double)lr * sqrt(1.0 - pow((double)beta2, (double)opt_step)) / (1.0 - pow((double)beta1, (double)opt_step));
        AdamArgs a;
        a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.weight_decay = weight_decay; a.step_size = (float)ss;
        a.grad_scale = grad_scale;
        pa.adam[0] = a;
        a.weight_decay = 0.f;
        pa.adam[1] = a;
        pa.adam_dst = e->adam_state(ws);
    }
    CK(step_prologue(pa, st));
    if (mode == 2 || force_launches) {
        e->dyn = true;
        int r = MB_OK;
        for (int sg = 0; sg < nseg && r == MB_OK; ++sg) {
            r = enqueue(sg, logits, loss, loss_run, m, v, loss_scale, st);
            if (r == MB_OK) r = between(sg, st);
        }
        e->dyn = false;
        flags.ok = r == MB_OK;
        return r;
    }
    StepGraph* g = nullptr;
    for (auto& x : e->graphs)
        if (x.B == B && x.L == L && x.with_opt == (m != nullptr) && x.overwrite == ow && x.logits == logits && x.loss == loss && x.loss_run == loss_run &&
            x.m == m && x.v == v && x.loss_scale == loss_scale && x.st == st && x.nseg == nseg && x.variant == variant && x.tag == tag) { g = &x; break; }
    if (!g) {
        if (e->graphs.size() >= 32) {          // callers that keep changing output pointers: do not grow without bound
            e->graphs.front().destroy();
            e->graphs.erase(e->graphs.begin());
        }
        StepGraph ng = {B, L, m != nullptr, ow, logits, loss, loss_run, m, v, loss_scale, st, nseg, variant, tag, {}, {}};
        hipStream_t cs = nullptr;
        CK(e->capture_stream(&cs));
        for (int sg = 0; sg < nseg; ++sg) {
            hipGraph_t gr = nullptr;
            hipGraphExec_t ex = nullptr;
            CK((int)hipStreamBeginCapture(cs, hipStreamCaptureModeRelaxed));
            e->dyn = true; e->capturing = true;
            const int r = enqueue(sg, logits, loss, loss_run, m, v, loss_scale, cs);
            e->dyn = false; e->capturing = false;
            const int r2 = (int)hipStreamEndCapture(cs, &gr);
            if (r || r2) { mb::ck_trace(r ? "enqueue(segment) inside the capture" : "hipStreamEndCapture", __FILE__, __LINE__, r ? r : r2);
                           if (gr) hipGraphDestroy(gr); ng.destroy(); return r ? r : r2; }
            if (verify) { const int rv = verify(tag, nseg, sg, gr); if (rv) { mb::ck_trace("finish(segment graph)", __FILE__, __LINE__, rv); hipGraphDestroy(gr); ng.destroy(); return rv; } }
            const int r3 = (int)hipGraphInstantiate(&ex, gr, nullptr, nullptr, 0);
            if (r3) { mb::ck_trace("hipGraphInstantiate", __FILE__, __LINE__, r3); hipGraphDestroy(gr); ng.destroy(); return r3; }
            ng.graph.push_back(gr); ng.exec.push_back(ex);
        }
        e->graphs.push_back(ng);
        g = &e->graphs.back();
        ++e->graph_captures;
    }
    for (int sg = 0; sg < nseg; ++sg) {
        CK((int)hipGraphLaunch(g->exec[sg], st));
        CK(between(sg, st));
    }
    ++e->graph_launches;
    flags.ok = true;
    return MB_OK;
}

// ------------------------------------------------------------------------------------------------ MAG operator
// rows [T, Tp) of MAG's k-major GEMM operands (packed modalities, the three dZ): cleared by the engines whenever the token count
// changes (never inside a captured step), instead of by a launch in every forward and every backward
inline int mag_clear_pad_rows(int dtype, char* ws, const MagWs& w, int T, int H, hipStream_t st) {
    const size_t Tp = align_up((size_t)T, 64), es = esize(dtype);
    if (Tp == (size_t)T) return MB_OK;
    auto zp = [&](size_t off, size_t cols) { return (int)hipMemsetAsync(ws + off + (size_t)T * cols * es, 0, (Tp - T) * cols * es, st); };
    CK(zp(w.vp, w.Vp)); CK(zp(w.ap, w.Ap));
    CK(zp(w.dZe, 2 * (size_t)H)); CK(zp(w.dZv, 2 * (size_t)H)); CK(zp(w.dZa, 2 * (size_t)H));
    return MB_OK;
}
inline int mag_fwd_impl(int dtype, const void* text, const float* visual, const float* acoustic, const float* W_hv,
                 const float* b_hv, const float* W_ha, const float* b_ha, const float* W_v, const float* b_v,
                 const float* W_a, const float* b_a, const float* ln_w, const float* ln_b, float ln_eps, float beta_shift,
                 DropKey drop, void* out, char* ws, const MagWs& w, int T, int H, int V, int A, bool repack, hipStream_t st,
                 bool pads_clean = false, bool packed = false) {
    // packed: the step prologue already wrote this batch's packed modality operands (ws + w.vp / w.ap)
    // pads_clean: the caller keeps rows [T, Tp) of the k-major operands zero itself (the engines: mag_clear_pad_rows whenever the
    // token count changes -- no kernel ever writes those rows); the stand-alone operator clears them on every call
    MagDims d = {T, H, V, A, w.Vp, w.Ap};
    if (repack) CK(mag_pack_weights(dtype, W_hv, W_ha, W_v, W_a, ws + w.We, ws + w.Wv, ws + w.Wa, d, st));
    if (!packed) {
        CK(pack_pad(dtype, visual, V, ws + w.vp, w.Vp, T, st));
        CK(pack_pad(dtype, acoustic, A, ws + w.ap, w.Ap, T, st));
    }
    if (!pads_clean) {
        const int Tp = (int)align_up((size_t)T, 64);
        const size_t es = esize(dtype);
        if (Tp > T) {
            ZeroRanges z = {};
            z.add(ws + w.vp + (size_t)T * w.Vp * es, (size_t)(Tp - T) * w.Vp * es);
            z.add(ws + w.ap + (size_t)T * w.Ap * es, (size_t)(Tp - T) * w.Ap * es);
            CK(zero_fill_ranges(z, st));
        }
    }
    CK(gemm(dtype, GEMM_NT, EPI_BIAS, T, 2 * H, H, text, H, ws + w.We, H, ws + w.Ze, 2 * H, nullptr, nullptr, nullptr,
            nullptr, 0, kNoDrop, 1, 0, st));
    CK(gemm(dtype, GEMM_NT, EPI_BIAS, T, 2 * H, w.Vp, ws + w.vp, w.Vp, ws + w.Wv, w.Vp, ws + w.Zv, 2 * H, nullptr, nullptr,
            nullptr, nullptr, 0, kNoDrop, 1, 0, st));
    CK(gemm(dtype, GEMM_NT, EPI_BIAS, T, 2 * H, w.Ap, ws + w.ap, w.Ap, ws + w.Wa, w.Ap, ws + w.Za, 2 * H, nullptr, nullptr,
            nullptr, nullptr, 0, kNoDrop, 1, 0, st));
    CK(mag_gate_forward(dtype, text, ws + w.Ze, ws + w.Zv, ws + w.Za, b_hv, b_ha, b_v, b_a, ln_w, ln_b, ln_eps, beta_shift,
                        out, (float*)(ws + w.mean), (float*)(ws + w.rstd), d, drop, st));
    return MB_OK;
}

inline int mag_bwd_impl(int dtype, const void* d_out, const void* text, const float* b_hv, const float* b_ha, const float* b_v,
                 const float* b_a, const float* ln_w, float beta_shift, DropKey drop, char* ws, const MagWs& w, void* d_text,
                 float* d_visual, float* d_acoustic, float* dW_hv, float* db_hv, float* dW_ha, float* db_ha, float* dW_v,
                 float* db_v, float* dW_a, float* db_a, float* dln_w, float* dln_b, int T, int H, int V, int A,
                 bool text_padded, hipStream_t st, GradAcc acc = {}, bool pads_clean = false, float* part_a = nullptr,
                 float* part_b = nullptr, int* part_nblk = nullptr, bool dw_zero = false) {
    // dw_zero: the caller knows the four weight-gradient tensors hold zeros (the engines' store path): plain stores, no read-modify-write
    MagDims d = {T, H, V, A, w.Vp, w.Ap};
    const int Tp = (int)align_up((size_t)T, 64);      // zero-padded token rows of the workspace operands
    const size_t es = esize(dtype);
    const int Kt = text_padded ? Tp : T;
    // The weight gradients as ONE grouped launch (like a layer's four): alone, the modality problems are 24 / 48 tiles whose duration
    // is the K = T loop latency (three launches of ~34 us each).
    //  * direct (default): six problems of H output rows each -- (text | visual | acoustic operand) x (gate half | projection half of
    //    the dZ columns) -- that store straight into dW_hv / dW_ha / dW_v / dW_a: column offsets V / A inside rows 815 / 842 floats
    //    wide, only the V / A real columns of the padded modality operands (GemmArgs::cvalid: dword stores, no alignment needed).
    //    No packed accumulators, no unpack launch (round 4: 7.4 us + 12 MB of traffic per step).
    //  * MB_MAG_WGRAD_DIRECT=0: three problems into packed fp32 scratch + mag_unpack_wgrads (rounds 2-3).
    const char* dv = getenv("MB_MAG_WGRAD_DIRECT");
    const int direct_env = dv ? atoi(dv) : 1;
    const char* dZe = ws + w.dZe; const char* dZv = ws + w.dZv; const char* dZa = ws + w.dZa;
    GemmArgs wg[6];
    int nwg = 3;
    wg[0] = wgrad_args(2 * H, H, Kt, dZe, 2 * H, text, H, (float*)(ws + w.dWe), H);
    wg[1] = wgrad_args(2 * H, w.Vp, Tp, dZv, 2 * H, ws + w.vp, w.Vp, (float*)(ws + w.dWv), w.Vp);
    wg[2] = wgrad_args(2 * H, w.Ap, Tp, dZa, 2 * H, ws + w.ap, w.Ap, (float*)(ws + w.dWa), w.Ap);
    const int wtile = (w.Vp % 128 == 0 && w.Ap % 128 == 0 && H % 128 == 0 && mag_wgrad_tile() == 128 && gemm_grouped_tn_ok(dtype, wg, 3, 128)) ? 128 : 64;
    const bool grouped = text_padded && gemm_grouped_tn_ok(dtype, wg, 3, wtile);
    const bool direct = grouped && direct_env != 0;
    if (direct) {
        const size_t half = (size_t)H * es;          // the projection half of a dZ row starts H columns in
        wg[0] = wgrad_args(H, H, Kt, dZe, 2 * H, text, H, dW_hv + V, V + H);                 wg[0].cvalid = H;
        wg[1] = wgrad_args(H, H, Kt, dZe + half, 2 * H, text, H, dW_ha + A, A + H);          wg[1].cvalid = H;
        wg[2] = wgrad_args(H, w.Vp, Tp, dZv, 2 * H, ws + w.vp, w.Vp, dW_hv, V + H);          wg[2].cvalid = V;
        wg[3] = wgrad_args(H, w.Vp, Tp, dZv + half, 2 * H, ws + w.vp, w.Vp, dW_v, V);        wg[3].cvalid = V;
        wg[4] = wgrad_args(H, w.Ap, Tp, dZa, 2 * H, ws + w.ap, w.Ap, dW_ha, A + H);          wg[4].cvalid = A;
        wg[5] = wgrad_args(H, w.Ap, Tp, dZa + half, 2 * H, ws + w.ap, w.Ap, dW_a, A);        wg[5].cvalid = A;
        nwg = 6;
        if (!gemm_grouped_tn_ok(dtype, wg, nwg, wtile)) return MB_ERR_SHAPE;
    }
    {
        // one launch clears the pad rows of this call's k-major operands and (ungrouped path) the packed weight-gradient accumulators
        ZeroRanges z = {};
        if (Tp > T && !pads_clean) {
            z.add(ws + w.dZe + (size_t)T * 2 * H * es, (size_t)(Tp - T) * 2 * H * es);
            z.add(ws + w.dZv + (size_t)T * 2 * H * es, (size_t)(Tp - T) * 2 * H * es);
            z.add(ws + w.dZa + (size_t)T * 2 * H * es, (size_t)(Tp - T) * 2 * H * es);
        }
        if (!grouped) {
            z.add(ws + w.dWe, (size_t)2 * H * H * 4);
            z.add(ws + w.dWv, (size_t)2 * H * w.Vp * 4);
            z.add(ws + w.dWa, (size_t)2 * H * w.Ap * 4);
        }
        if (z.n) CK(zero_fill_ranges(z, st));
    }
    CK(mag_gate_backward(dtype, d_out, text, ws + w.Ze, ws + w.Zv, ws + w.Za, b_hv, b_ha, b_v, b_a, ln_w,
                         (const float*)(ws + w.mean), (const float*)(ws + w.rstd), beta_shift, ws + w.dep, ws + w.dZe,
                         ws + w.dZv, ws + w.dZa, db_hv, db_ha, db_v, db_a, dln_w, dln_b, d, drop, st, acc, part_a, part_b, part_nblk));
    if (grouped) {
        for (int i = 0; i < nwg; ++i) wg[i].overwrite = direct ? (dw_zero ? 1 : 0) : 1;      // (the packed accumulators are always stored)
        CK(gemm_grouped_tn_launch(dtype, wg, nwg, wtile, st, wtile == 64 ? mag_wgrad_stages() : 0));
    } else {
        CK(wgrad(dtype, 2 * H, H, text_padded ? Tp : T, ws + w.dZe, 2 * H, text, H, (float*)(ws + w.dWe), H, st));
        CK(wgrad(dtype, 2 * H, w.Vp, Tp, ws + w.dZv, 2 * H, ws + w.vp, w.Vp, (float*)(ws + w.dWv), w.Vp, st));
        CK(wgrad(dtype, 2 * H, w.Ap, Tp, ws + w.dZa, 2 * H, ws + w.ap, w.Ap, (float*)(ws + w.dWa), w.Ap, st));
    }
    if (!direct)
        CK(mag_unpack_wgrads((const float*)(ws + w.dWe), (const float*)(ws + w.dWv), (const float*)(ws + w.dWa), dW_hv, dW_ha,
                             dW_v, dW_a, d, st));
    // d_text = dZe . We + (ds + d||e|| term)
    CK(gemm(dtype, GEMM_NN, EPI_ADD_RES, T, H, 2 * H, ws + w.dZe, 2 * H, ws + w.We, H, d_text, H, nullptr, nullptr, nullptr,
            ws + w.dep, H, kNoDrop, 1, 0, st));
    if (d_visual) {
        CK(gemm(dtype, GEMM_NN, EPI_BIAS_F32, T, w.Vp, 2 * H, ws + w.dZv, 2 * H, ws + w.Wv, w.Vp, nullptr, w.Vp, nullptr,
                (float*)(ws + w.dvp), nullptr, nullptr, 0, kNoDrop, 1, 0, st));
        CK((int)hipMemcpy2DAsync(d_visual, (size_t)V * 4, ws + w.dvp, (size_t)w.Vp * 4, (size_t)V * 4, T,
                                 hipMemcpyDeviceToDevice, st));
    }
    if (d_acoustic) {
        CK(gemm(dtype, GEMM_NN, EPI_BIAS_F32, T, w.Ap, 2 * H, ws + w.dZa, 2 * H, ws + w.Wa, w.Ap, nullptr, w.Ap, nullptr,
                (float*)(ws + w.dap), nullptr, nullptr, 0, kNoDrop, 1, 0, st));
        CK((int)hipMemcpy2DAsync(d_acoustic, (size_t)A * 4, ws + w.dap, (size_t)w.Ap * 4, (size_t)A * 4, T,
                                 hipMemcpyDeviceToDevice, st));
    }
    return MB_OK;
}


}  // namespace
