// Gradient exchange of the data-parallel training step, issued from C on a side HIP stream.
//
// NEW relative to the reference (single device: /root/reference/global_configs.py:4,7; multimodal_driver.py:21 imports a
// DistributedSampler it never uses).  north_star: "data-parallel fine-tuning shards minibatches across the 8 GPUs of one node with a
// single RCCL all-reduce of gradients over xGMI per step, overlapped with the backward on a side HIP stream".
//
// One logical all-reduce(sum) of the flat fp32 gradient buffer per optimizer step, cut where the backward finishes ranges of it:
//   * the layers' GEMM weight gradients (77 % of the bytes) in chunks of MB_DP_CHUNK layers, each issued the moment its segment of
//     the backward has been enqueued (event on the compute stream -> the comm stream waits -> ncclAllReduce in place);
//   * the tail (pooler, embeddings, MAG, classifier, every bias / LayerNorm) after the last backward stage.  The word-embedding
//     table -- 30,522 x 768 fp32 = 94 MB of which at most B*L rows per rank are non-zero -- moves ROW-WISE: every rank all-gathers
//     the ids it touched and those rows (world x 7.4 MB instead of a 94 MB dense piece that would be produced last and fully exposed).
// xGMI is point-to-point (7 links per GPU): a few large contiguous pieces let RCCL's rings use every link; no bucket copies exist
// because the flat layout already makes each piece one contiguous range.  1/world is folded into the AdamW kernel.
//
// RCCL is loaded at run time (dlopen): the library has no link-time dependency on it, a host process that already carries an RCCL
// (PyTorch does) shares that copy, and single-GPU users never load it.  The callback backend exists for the two-rank equality
// tests, which run two ranks on ONE GPU (RCCL refuses that) over gloo.
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <rccl/rccl.h>
#include "comm.h"

using namespace mb;

#define CK(x) do { int _e = (x); if (_e) { mb::ck_trace(#x, __FILE__, __LINE__, _e); return _e; } } while (0)

namespace {

enum { MB_ERR_NCCL_BASE = 2000 };

// ------------------------------------------------------------------------------------------------ RCCL, resolved at run time
struct Rccl {
    void* h = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclReduceScatter) ReduceScatter = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    char err[256] = {0};
};
Rccl g_rccl;
std::once_flag g_rccl_once;
char g_last_error[512] = {0};

void load_rccl() {
    Rccl& r = g_rccl;
    // a copy that is already in the process first (torch/lib/librccl.so has no SONAME: it is known as "librccl.so"), then the ROCm install
    const char* env = getenv("MB_RCCL_PATH");
    const char* names[] = {env, "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
        if (!n || !*n) continue;
        r.h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
        if (r.h) break;
    }
    for (int i = 0; i < 4 && !r.h; ++i) {
        if (!names[i] || !*names[i]) continue;
        r.h = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
    }
    if (!r.h) { snprintf(r.err, sizeof r.err, "librccl.so not found (%s)", dlerror()); return; }
#define SYM(f) r.f = (decltype(r.f))dlsym(r.h, "nccl" #f); if (!r.f) { snprintf(r.err, sizeof r.err, "nccl" #f " missing in librccl"); return; }
    SYM(GetUniqueId) SYM(CommInitRank) SYM(CommDestroy) SYM(AllReduce) SYM(AllGather) SYM(ReduceScatter) SYM(GroupStart) SYM(GroupEnd) SYM(GetErrorString)
#undef SYM
}
int rccl_ready() {
    std::call_once(g_rccl_once, load_rccl);
    if (g_rccl.err[0]) { snprintf(g_last_error, sizeof g_last_error, "%s", g_rccl.err); return MB_ERR_COMM; }
    return MB_OK;
}
int nccl_rc(ncclResult_t r, const char* what) {
    if (r == ncclSuccess) return MB_OK;
    snprintf(g_last_error, sizeof g_last_error, "%s: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
    return MB_ERR_NCCL_BASE + (int)r;
}

// ------------------------------------------------------------------------------------------------ row exchange kernels
// slot[r][id] (int32, -1 between exchanges): the position inside rank r's send buffer that carries row `id`, if rank r touched it.
// A rank sends each touched row ONCE (the position that wins the atomicMax owns it); after the all-gather every rank fills the
// other ranks' slot tables from the gathered ids, and the lowest rank that touched a row sums the contributions in rank order and
// STORES the result: no atomics, no clearing pass, and every rank performs the same additions in the same order (the replicas'
// gradients stay bit-identical).
__global__ void rows_mark_kernel(const int64_t* __restrict__ ids, int T, int* __restrict__ slot_mine) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < T) atomicMax(&slot_mine[(int)ids[t]], t);
}
__global__ void rows_pack_kernel(const int64_t* __restrict__ ids, int T, const int* __restrict__ slot_mine, const float* __restrict__ table,
                                 int H, int* __restrict__ send_ids, float* __restrict__ send_rows) {
    const int t = blockIdx.x;
    int id = -1;
    if (t < T) {
        const int cand = (int)ids[t];
        if (slot_mine[cand] == t) id = cand;
    }
    if (threadIdx.x == 0) send_ids[t] = id;
    if (id < 0) return;
    const f32x4* src = (const f32x4*)(table + (size_t)id * H);
    f32x4* dst = (f32x4*)(send_rows + (size_t)t * H);
    for (int i = threadIdx.x; i < H / 4; i += blockDim.x) dst[i] = src[i];
}
__global__ void rows_index_kernel(const int* __restrict__ recv_ids, int total, int cap, int vocab, int* __restrict__ slot, int set) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int id = recv_ids[i];
    if (id >= 0) slot[(size_t)(i / cap) * vocab + id] = set ? i % cap : -1;
}
__global__ void rows_combine_kernel(const int* __restrict__ recv_ids, const float* __restrict__ recv_rows, int world, int cap, int vocab,
                                    int H, const int* __restrict__ slot, float* __restrict__ table) {
    const int r = blockIdx.x / cap, t = blockIdx.x % cap;
    const int id = recv_ids[(size_t)r * cap + t];
    if (id < 0) return;
    for (int q = 0; q < r; ++q)
        if (slot[(size_t)q * vocab + id] >= 0) return;          // a lower rank touched this row too: it does the sum
    for (int i = threadIdx.x; i < H / 4; i += blockDim.x) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int q = r; q < world; ++q) {
            const int s = slot[(size_t)q * vocab + id];
            if (s >= 0) acc += ((const f32x4*)(recv_rows + ((size_t)q * cap + s) * H))[i];
        }
        ((f32x4*)(table + (size_t)id * H))[i] = acc;
    }
}

int all_gather(mb_comm* c, void* buf, size_t bytes_per_rank, hipStream_t s) {
    if (c->nccl) {
        const char* send = (const char*)buf + (size_t)c->rank * bytes_per_rank;       // in place: the rank's piece sits where it will be received
        return nccl_rc(g_rccl.AllGather(send, buf, bytes_per_rank, ncclInt8, (ncclComm_t)c->nccl, s), "ncclAllGather");
    }
    if (c->ag_cb) return c->ag_cb(c->ctx, buf, bytes_per_rank, (void*)s);
    return MB_ERR_COMM;
}

}  // namespace

namespace mb {

// MB_DP_SYNTH_GBPS=r (measurement only, one-rank groups: profiles/r06_dp_chunks.txt): behind every all-reduce piece the comm stream runs a
// copy of the piece onto itself by EIGHT workgroups, repeated until it has lasted bytes / r GB/s -- a stand-in for the time a ring
// all-reduce spends on the xGMI links (8 GPUs, 7 links of ~153 GB/s: ~300 GB/s of algorithmic bandwidth), with the memory traffic of one,
// so that the trade between seam cost (more, smaller pieces) and exposed tail (fewer, larger) shows on one GPU.  Results unchanged.
__global__ void __launch_bounds__(256) dp_synth_load_kernel(float* g, size_t n4, long long ticks) {
    const long long t0 = wall_clock64();           // 100 MHz
    f32x4* p = (f32x4*)g;
    do {
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) p[i] = p[i];
    } while (wall_clock64() - t0 < ticks);
}
static int synth_gbps() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("MB_DP_SYNTH_GBPS"); v = e ? atoi(e) : 0; if (v < 0) v = 0;
        if (v > 0) fprintf(stderr, "[magbert] WARNING: MB_DP_SYNTH_GBPS=%d is a measurement switch -- every all-reduce piece is followed by a synthetic "
                                   "load of bytes / %d GB/s on the comm stream\n", v, v);
    }
    return v;
}

int comm_all_reduce(mb_comm* c, float* g, size_t count, hipStream_t s) {
    if (!c || !g) return MB_ERR_ARG;
    if (count == 0) return MB_OK;
    ++c->pieces; c->bytes_reduced += count * (c->wire == DT_BF16 ? 2 : 4);
    if (const int r = synth_gbps()) {
        if (c->world == 1 && count % 4 == 0) {
            const double us = (double)count * (c->wire == DT_BF16 ? 2.0 : 4.0) / ((double)r * 1e3);
            hipLaunchKernelGGL(dp_synth_load_kernel, dim3(8), dim3(256), 0, s, g, count / 4, (long long)(us * 100.0));
        }
    }
    if (c->wire == DT_BF16) {
        // bf16 wire: every rank rounds its gradients once, the sum travels and is accumulated in bf16, moments and parameters stay
        // fp32.  Halves the bytes on the links (two GPUs share ONE xGMI link: the fp32 exchange lasts as long as the backward).
        if (!c->scratch || count > c->n_params) return MB_ERR_ARG;
        char* stage = c->scratch + c->off_stage;
        CK(convert(DT_BF16, g, stage, count, s));
        if (c->nccl) CK(nccl_rc(g_rccl.AllReduce(stage, stage, count, ncclBfloat16, ncclSum, (ncclComm_t)c->nccl, s), "ncclAllReduce"));
        else if (c->ar_cb) CK(c->ar_cb(c->ctx, stage, count, DT_BF16, (void*)s));
        else return MB_ERR_COMM;
        return widen(DT_BF16, stage, g, count, s);
    }
    if (c->nccl) return nccl_rc(g_rccl.AllReduce(g, g, count, ncclFloat, ncclSum, (ncclComm_t)c->nccl, s), "ncclAllReduce");
    if (c->ar_cb) return c->ar_cb(c->ctx, g, count, DT_F32, (void*)s);
    return MB_ERR_COMM;
}

// sum over the ranks of g[0, world * per), delivered slice-wise: afterwards g[rank * per, (rank + 1) * per) holds the sum on this rank
// (the other slices are dead).  In place.  The callback backend has no reduce-scatter: its all-reduce delivers a superset.
static int comm_reduce_scatter(mb_comm* c, float* g, size_t per, hipStream_t s) {
    if (per == 0) return MB_OK;
    const size_t count = per * (size_t)c->world;
    ++c->pieces; c->bytes_reduced += count * (c->wire == DT_BF16 ? 2 : 4) / 2;      // (half an all-reduce's traffic)
    if (c->wire == DT_BF16) {
        if (!c->scratch || count > c->n_params) return MB_ERR_ARG;
        char* stage = c->scratch + c->off_stage;
        char* mine = stage + (size_t)c->rank * per * 2;
        CK(convert(DT_BF16, g, stage, count, s));
        if (c->nccl) CK(nccl_rc(g_rccl.ReduceScatter(stage, mine, per, ncclBfloat16, ncclSum, (ncclComm_t)c->nccl, s), "ncclReduceScatter"));
        else if (c->ar_cb) CK(c->ar_cb(c->ctx, stage, count, DT_BF16, (void*)s));
        else return MB_ERR_COMM;
        return widen(DT_BF16, mine, g + (size_t)c->rank * per, per, s);
    }
    if (c->nccl) return nccl_rc(g_rccl.ReduceScatter(g, g + (size_t)c->rank * per, per, ncclFloat, ncclSum, (ncclComm_t)c->nccl, s), "ncclReduceScatter");
    if (c->ar_cb) return c->ar_cb(c->ctx, g, count, DT_F32, (void*)s);
    return MB_ERR_COMM;
}
// in-place all-gather of the `world` slices of `per` elements (of `es` bytes) that start at element `b` of `base`
static int comm_gather_slices(mb_comm* c, char* base, int es, size_t b, size_t per, hipStream_t s) {
    if (per == 0) return MB_OK;
    c->bytes_gathered += per * (size_t)es * (size_t)c->world;
    return all_gather(c, base + b * (size_t)es, per * (size_t)es, s);
}

int comm_exchange_rows(mb_comm* c, float* table, const int64_t* ids, int T, hipStream_t s) {
    if (!c || !c->rows_ready || !table || !ids || T < 1 || T > c->cap || (c->H & 3)) return MB_ERR_ARG;
    const int world = c->world, cap = c->cap, vocab = c->vocab, H = c->H;
    int* slot = (int*)(c->scratch + c->off_slot);
    int* rids = (int*)(c->scratch + c->off_ids);
    float* rrows = (float*)(c->scratch + c->off_rows);
    int* slot_mine = slot + (size_t)c->rank * vocab;
    rows_mark_kernel<<<(T + 255) / 256, 256, 0, s>>>(ids, T, slot_mine);
    rows_pack_kernel<<<cap, 192, 0, s>>>(ids, T, slot_mine, table, H, rids + (size_t)c->rank * cap, rrows + (size_t)c->rank * cap * H);
    CK((int)hipGetLastError());
    CK(all_gather(c, rids, (size_t)cap * sizeof(int), s));
    CK(all_gather(c, rrows, (size_t)cap * H * sizeof(float), s));
    const int total = world * cap;
    rows_index_kernel<<<(total + 255) / 256, 256, 0, s>>>(rids, total, cap, vocab, slot, 1);
    rows_combine_kernel<<<total, 192, 0, s>>>(rids, rrows, world, cap, vocab, H, slot, table);
    rows_index_kernel<<<(total + 255) / 256, 256, 0, s>>>(rids, total, cap, vocab, slot, 0);      // back to all -1 for the next step
    c->bytes_reduced += (size_t)cap * (sizeof(int) + (size_t)H * sizeof(float));
    c->pieces += 2;
    return (int)hipGetLastError();
}

// the comm stream picks up after everything enqueued on `st` so far (segment `seg` of the step)
static int fork_to_comm(mb_comm* c, int seg, hipStream_t st) {
    hipEvent_t ev = c->fork_ev[(size_t)seg % c->fork_ev.size()];
    if (c->event_mode < 2) CK((int)hipEventRecord(ev, st));      // (modes 2, 3: recorded by the segment itself, dp_segment_end)
    static int dbg = -1;
    if (dbg < 0) {
        const char* v = getenv("MB_DP_DEBUG"); dbg = v ? atoi(v) : 0;
        if (dbg == 4) fprintf(stderr, "[magbert] WARNING: MB_DP_DEBUG=4 is a test switch -- the gradient exchange is NOT ordered behind the backward; "
                                      "training results are wrong by design\n");
    }
    if (dbg == 4) return MB_OK;          // NEGATIVE CONTROL of the equality tests: the hand-off is dropped, the exchange races the backward
    CK((int)hipStreamWaitEvent(c->cs, ev, 0));
    return MB_OK;
}
// MB_DP_TEST_DELAY_US (tests only): every backward segment starts with a kernel that spins this long, so that the host has issued
// the segment's collectives long before the segment's gradients exist -- a missing compute -> comm dependency then fails the
// equality tests deterministically instead of hiding behind a slow host (tests/test_dp_gpu.py)
__global__ void dp_test_delay_kernel(long long ticks) {
    const long long t0 = wall_clock64();           // 100 MHz
    while (wall_clock64() - t0 < ticks) {}
}
static int test_delay_us() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("MB_DP_TEST_DELAY_US"); v = e ? atoi(e) : 0; if (v < 0) v = 0;
        if (v > 0) fprintf(stderr, "[magbert] WARNING: MB_DP_TEST_DELAY_US=%d is a test switch -- every backward segment of the data-parallel step "
                                   "starts with a %d us spin kernel (baked into the captured graphs)\n", v, v);
    }
    return v;
}
static bool is_capturing(hipStream_t st) {
    hipStreamCaptureStatus cst = hipStreamCaptureStatusNone;
    return hipStreamIsCapturing(st, &cst) == hipSuccess && cst == hipStreamCaptureStatusActive;
}
// which segments end with a hand-off to the comm stream: the backward segments; under the sharded update the two optimizer segments too
// (their slices go out by all-gather)
static bool hands_off(const mb_comm* c, int nb, int seg) { return seg < nb || (c->shard && seg < nb + 2); }
int dp_segment_end(mb_comm* c, int nb, int seg, hipStream_t st) {
    if (seg < c->nf) return MB_OK;          // a forward-only piece hands nothing over
    if (c->event_mode < 2 || !hands_off(c, nb, seg - c->nf)) return MB_OK;
    // captured: nothing here -- a plain hipEventRecord inside a capture adds NO node (it only orders captured work), and the external
    // form (hipEventRecordWithFlags) is refused by the 7.0 runtime a PyTorch process maps; dp_finish_segment_graph appends the
    // event-record node to the captured graph instead
    if (is_capturing(st)) return MB_OK;
    CK((int)hipEventRecord(c->fork_ev[(size_t)seg % c->fork_ev.size()], st));
    return MB_OK;
}
int dp_segment_begin(mb_comm* c, int nb, int seg_abs, hipStream_t st) {
    if (seg_abs < c->nf) return MB_OK;
    const int seg = seg_abs - c->nf;
    if (seg < nb && test_delay_us() > 0) {
        dp_test_delay_kernel<<<1, 1, 0, st>>>((long long)test_delay_us() * 100);
        CK((int)hipGetLastError());
    }
    if (c->event_mode < 3 || seg < nb) return MB_OK;
    if (is_capturing(st)) return MB_OK;          // (dp_finish_segment_graph puts the wait node in front of the captured sequence)
    CK((int)hipStreamWaitEvent(st, seg == nb ? (nb > 1 ? c->ev_layers : c->ev_tail) : c->ev_tail, 0));
    return MB_OK;
}
// After the capture of segment `seg` (train_step_impl's hook; tag = the mb_comm, nseg = nb + 2): the hand-off events become NODES of
// the segment's graph -- event modes 2 / 3: a backward segment ENDS with an event-record node of its fork event (the comm stream's
// hipStreamWaitEvent, issued right behind the graph launch, then waits for this replay's record); mode 3: an optimizer segment
// STARTS with a wait-event node on the comm stream's "pieces enqueued" event.  The graph stays a linear chain.  Then the result is
// checked: exactly one such node, at the right end (tools/event_capture_probe.cpp is the stand-alone demonstration that these
// nodes order a replay against another stream, and that a record issued inside the capture does not).
int dp_finish_segment_graph(const void* tag, int nseg, int seg_abs, hipGraph_t graph) {
    mb_comm* c = (mb_comm*)tag;
    if (!c || c->event_mode < 2 || seg_abs < c->nf) return MB_OK;
    const int nb = nseg - 2 - c->nf, seg = seg_abs - c->nf;
    auto nodes_of = [&](std::vector<hipGraphNode_t>& v) -> int {
        size_t n = 0;
        CK((int)hipGraphGetNodes(graph, nullptr, &n));
        v.resize(n);
        if (n) CK((int)hipGraphGetNodes(graph, v.data(), &n));
        return MB_OK;
    };
    std::vector<hipGraphNode_t> nodes;
    CK(nodes_of(nodes));
    if (hands_off(c, nb, seg)) {
        std::vector<hipGraphNode_t> leaves;
        for (auto nd : nodes) { size_t nout = 0; CK((int)hipGraphNodeGetDependentNodes(nd, nullptr, &nout)); if (nout == 0) leaves.push_back(nd); }
        hipGraphNode_t rec = nullptr;
        CK((int)hipGraphAddEventRecordNode(&rec, graph, leaves.data(), leaves.size(), c->fork_ev[(size_t)seg_abs % c->fork_ev.size()]));
    }
    if (seg >= nb && c->event_mode >= 3) {
        size_t nr = 0;
        CK((int)hipGraphGetRootNodes(graph, nullptr, &nr));
        std::vector<hipGraphNode_t> roots(nr);
        if (nr) CK((int)hipGraphGetRootNodes(graph, roots.data(), &nr));
        hipGraphNode_t wn = nullptr;
        CK((int)hipGraphAddEventWaitNode(&wn, graph, nullptr, 0, seg == nb ? (nb > 1 ? c->ev_layers : c->ev_tail) : c->ev_tail));
        for (auto r : roots) CK((int)hipGraphAddDependencies(graph, &wn, &r, 1));
    }
    // verification
    CK(nodes_of(nodes));
    int records = 0, waits = 0, record_is_leaf = 0, wait_is_root = 0, leaves = 0, roots = 0;
    for (auto nd : nodes) {
        hipGraphNodeType t;
        CK((int)hipGraphNodeGetType(nd, &t));
        size_t nout = 0, nin = 0;
        CK((int)hipGraphNodeGetDependentNodes(nd, nullptr, &nout));
        CK((int)hipGraphNodeGetDependencies(nd, nullptr, &nin));
        leaves += nout == 0; roots += nin == 0;
        if (t == hipGraphNodeTypeEventRecord) { ++records; record_is_leaf += nout == 0; }
        if (t == hipGraphNodeTypeWaitEvent) { ++waits; wait_is_root += nin == 0; }
    }
    const bool ok = (!hands_off(c, nb, seg) || (records == 1 && record_is_leaf == 1 && leaves == 1)) &&
                    (seg < nb || c->event_mode < 3 || (waits == 1 && wait_is_root == 1 && roots == 1));
    if (!ok) {
        snprintf(g_last_error, sizeof g_last_error, "data-parallel segment %d of %d: graph has %d event-record / %d wait-event nodes, %d leaves, "
                 "%d roots (event mode %d): the hand-off to the comm stream would not be ordered", seg_abs, nseg, records, waits, leaves, roots, c->event_mode);
        return MB_ERR_MODE;
    }
    return MB_OK;
}
// the compute stream waits for `ev` (recorded on the comm stream); with timing on, the stall is bracketed by two timing events
static int wait_timed(mb_comm* c, int k, hipEvent_t ev, hipStream_t st) {
    if (c->event_mode >= 3) return MB_OK;          // (the waiting segment's graph starts with the wait: dp_segment_begin)
    if (c->timing) CK((int)hipEventRecord(c->tev[2 * k], st));
    CK((int)hipStreamWaitEvent(st, ev, 0));
    if (c->timing) { CK((int)hipEventRecord(c->tev[2 * k + 1], st)); c->tev_used[k] = true; }
    return MB_OK;
}

std::vector<int> dp_chunk_plan(int n_layer) {
    std::vector<int> plan;
    if (const char* v = getenv("MB_DP_CHUNKS")) {
        int sum = 0;
        for (const char* p = v; *p;) {
            const int c = atoi(p);
            if (c < 1) { plan.clear(); break; }
            plan.push_back(c); sum += c;
            while (*p && *p != ',') ++p;
            if (*p == ',') ++p;
        }
        if (sum == n_layer && !plan.empty()) return plan;
        plan.clear();
    }
    if (const char* v = getenv("MB_DP_CHUNK")) {
        const int c = atoi(v);
        if (c > 0 && c <= n_layer && n_layer % c == 0) { plan.assign((size_t)(n_layer / c), c); return plan; }
    }
    int left = n_layer;
    while (left > 4 && left - 4 >= 4) { plan.push_back(4); left -= 4; }
    while (left > 2) { plan.push_back(2); left -= 2; }
    if (left > 0) plan.push_back(left);
    return plan;
}

int dp_between(mb_comm* c, const DpSpec& sp, float* G, int seg, hipStream_t st) {
    // MB_DP_DEBUG (measurement only -- the gradients are NOT exchanged): 1 = the segmented step alone (no events, no collectives),
    // 2 = events and waits but no collectives, 3 = everything except the row-wise exchange (the table is left as it is);
    // 4 = (tests' negative control) everything, but the comm stream does not wait for the backward segment (fork_to_comm)
    static int dbg = -1;
    if (dbg < 0) { const char* v = getenv("MB_DP_DEBUG"); dbg = v ? atoi(v) : 0; }
    if (dbg == 1) return MB_OK;
    const int nb = (int)sp.chunk.size();
    const int seg_abs = seg;
    if (seg_abs == 0) {
        c->pieces = 0; c->bytes_reduced = 0; c->bytes_gathered = 0; c->tev_used[0] = c->tev_used[1] = false;
        if (c->shard) c->shard_chunks.assign(sp.chunk.begin(), sp.chunk.begin() + sp.n_sharded);
    }
    if (seg_abs < c->nf) {
        // behind forward piece k (chunk nb-1-k): the NEXT piece's weights (chunk nb-2-k) come from the previous step's all-gather
        return dbg == 2 ? MB_OK : (int)hipStreamWaitEvent(st, c->ev_chunk[(size_t)(nb - 2 - seg_abs) % c->ev_chunk.size()], 0);
    }
    seg -= c->nf;
    auto piece = [&](size_t b, size_t e) -> int { return (dbg == 2 || e <= b) ? MB_OK : comm_all_reduce(c, G + b, e - b, c->cs); };
    // a chunk of layer GEMM weights: all-reduced, or -- sharded update -- reduce-scattered slice-wise (+ its replicated remainder)
    auto chunk_piece = [&](int k) -> int {
        const size_t b = sp.chunk[k].first, e = sp.chunk[k].second;
        if (k >= sp.n_sharded) return piece(b, e);
        const ShardSlice sl = dp_shard_slice(c, b, e);
        if (dbg != 2) CK(comm_reduce_scatter(c, G + b, sl.per, c->cs));
        return piece(sl.rem_b, sl.rem_e);
    };
    auto gather_chunk = [&](int k) -> int {
        if (k >= sp.n_sharded) return MB_OK;
        if (dbg != 2 && sp.gather_base) {
            const ShardSlice sl = dp_shard_slice(c, sp.chunk[k].first, sp.chunk[k].second);
            CK(comm_gather_slices(c, sp.gather_base, sp.gather_es, sp.chunk[k].first, sl.per, c->cs));
        }
        CK((int)hipEventRecord(c->ev_chunk[(size_t)k % c->ev_chunk.size()], c->cs));      // "chunk k's operands are everybody's"
        return MB_OK;
    };
    if (seg < nb - 1) {
        CK(fork_to_comm(c, seg_abs, st));
        return chunk_piece(seg);
    }
    if (seg == nb - 1) {
        CK((int)hipEventRecord(c->ev_layers, c->cs));          // every early piece is in front of this
        CK(fork_to_comm(c, seg_abs, st));
        CK(chunk_piece(seg));
        const size_t w0 = sp.word_off, w1 = sp.word_off + (size_t)sp.word_rows * sp.H;
        // (a batch beyond the agreed row capacity is an error, never a silent switch to the dense piece: the ranks must issue the
        //  same collectives in the same order)
        if (sp.word_rows > 0 && c->rows_ready && c->rowwise && sp.T > c->cap) return MB_ERR_SHAPE;
        if (sp.word_rows > 0 && c->rows_ready && c->rowwise && sp.word_rows == c->vocab && sp.H == c->H && w0 >= sp.tail_begin && w1 <= sp.tail_end) {
            CK(piece(sp.tail_begin, w0));
            if (dbg < 2) CK(comm_exchange_rows(c, G + w0, sp.ids, sp.T, c->cs));
            CK(piece(w1, sp.tail_end));
        } else {
            CK(piece(sp.tail_begin, sp.tail_end));
        }
        CK((int)hipEventRecord(c->ev_tail, c->cs));
        // the optimizer of the early ranges (next segment) needs those pieces only: it runs under the late pieces' exchange
        // (a one-segment backward has no early range: its next segment updates part of the late ones, so it waits for everything)
        return wait_timed(c, 0, nb > 1 ? c->ev_layers : c->ev_tail, st);
    }
    if (seg == nb) {
        // sharded update: the slices the first optimizer launch produced (the early chunks) go out while the second one runs, the
        // lowest layers first (the next forward needs them in that order)
        if (c->shard && nb > 1) {
            CK(fork_to_comm(c, seg_abs, st));
            for (int k = nb - 2; k >= 0; --k) CK(gather_chunk(k));
        }
        return wait_timed(c, 1, c->ev_tail, st);
    }
    if (seg == nb + 1 && c->shard) {
        CK(fork_to_comm(c, seg_abs, st));
        CK(gather_chunk(nb - 1));          // (several pieces: the last one is replicated -- nothing to gather)
        CK((int)hipEventRecord(c->ev_gather, c->cs));
        c->gather_pending = true;
    }
    return MB_OK;
}

// head of a data-parallel step.  cut = the step's forward is cut into pieces that wait for their own gathers (dp_between): nothing to
// wait for here; else (one piece, or a consumer outside the step: mb_comm_join) the stream waits for the last gather
int dp_step_begin(mb_comm* c, hipStream_t st, bool cut) {
    if (!c->gather_pending) return MB_OK;
    c->gather_pending = false;
    if (cut) return MB_OK;
    CK((int)hipStreamWaitEvent(st, c->ev_gather, 0));
    return MB_OK;
}

}  // namespace mb

// ================================================================================================ C ABI
extern "C" {

const char* mb_comm_last_error(void) { return g_last_error; }

int mb_comm_unique_id(void* id128) {
    if (!id128) return MB_ERR_ARG;
    CK(rccl_ready());
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    return nccl_rc(g_rccl.GetUniqueId((ncclUniqueId*)id128), "ncclGetUniqueId");
}

static int comm_common_init(mb_comm* c) {
    int least = 0, greatest = 0;
    CK((int)hipDeviceGetStreamPriorityRange(&least, &greatest));
    // Normal priority by default.  Round 4 gave the comm stream the highest priority ("the exchange's few workgroups go in front of
    // the backward's many": 3.71 vs 3.77 ms) -- measured while the hand-off events ordered nothing.  With real dependencies a
    // highest-priority comm stream inside a PyTorch process makes every hand-off cost ~1 ms (10.5 vs 3.67 ms per step; the torch-free
    // driver does not show it, and the runtime is not the reason: tools/stream_handoff_probe measures 12 us per hand-off under both;
    // profiles/r05_dp_comm_priority.txt).  MB_DP_COMM_PRIORITY=1 brings the high-priority stream back.
    const char* pv = getenv("MB_DP_COMM_PRIORITY");
    const int prio = (pv && atoi(pv) == 1) ? greatest : 0;
    CK((int)hipStreamCreateWithPriority(&c->cs, hipStreamNonBlocking, prio));
    if (const char* v = getenv("MB_DP_EVENT_MODE")) c->event_mode = atoi(v);
    const unsigned flags = hipEventDisableTiming | (c->event_mode == 1 ? hipEventReleaseToDevice : 0);
    c->fork_ev.assign(32, nullptr);
    for (auto& ev : c->fork_ev) CK((int)hipEventCreateWithFlags(&ev, flags));
    CK((int)hipEventCreateWithFlags(&c->ev_layers, flags));
    CK((int)hipEventCreateWithFlags(&c->ev_tail, flags));
    CK((int)hipEventCreateWithFlags(&c->ev_gather, flags));
    c->ev_chunk.assign(16, nullptr);
    for (auto& ev : c->ev_chunk) CK((int)hipEventCreateWithFlags(&ev, flags));
    for (auto& ev : c->tev) CK((int)hipEventCreate(&ev));
    return MB_OK;
}

int mb_comm_create_rccl(const void* id128, int rank, int world, mb_comm** out) {
    if (!id128 || !out || world < 1 || rank < 0 || rank >= world) return MB_ERR_ARG;
    CK(rccl_ready());
    mb_comm* c = new mb_comm();
    c->rank = rank; c->world = world;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    ncclComm_t comm = nullptr;
    int r = nccl_rc(g_rccl.CommInitRank(&comm, world, id, rank), "ncclCommInitRank");
    if (r == MB_OK) { c->nccl = comm; r = comm_common_init(c); }
    if (r != MB_OK) { mb_comm_destroy(c); return r; }
    *out = c;
    return MB_OK;
}

int mb_comm_create_callbacks(int rank, int world, mb_all_reduce_cb all_reduce, mb_all_gather_cb all_gather_, void* ctx, mb_comm** out) {
    if (!out || !all_reduce || world < 1 || rank < 0 || rank >= world) return MB_ERR_ARG;
    mb_comm* c = new mb_comm();
    c->rank = rank; c->world = world; c->ar_cb = all_reduce; c->ag_cb = all_gather_; c->ctx = ctx;
    const int r = comm_common_init(c);
    if (r != MB_OK) { mb_comm_destroy(c); return r; }
    *out = c;
    return MB_OK;
}

void mb_comm_destroy(mb_comm* c) {
    if (!c) return;
    if (c->cs) hipStreamSynchronize(c->cs);
    if (c->nccl && g_rccl.CommDestroy) g_rccl.CommDestroy((ncclComm_t)c->nccl);
    for (auto ev : c->fork_ev) if (ev) hipEventDestroy(ev);
    if (c->ev_layers) hipEventDestroy(c->ev_layers);
    if (c->ev_tail) hipEventDestroy(c->ev_tail);
    if (c->ev_gather) hipEventDestroy(c->ev_gather);
    for (auto ev : c->ev_chunk) if (ev) hipEventDestroy(ev);
    for (auto ev : c->tev) if (ev) hipEventDestroy(ev);
    if (c->cs) hipStreamDestroy(c->cs);
    delete c;
}

int mb_comm_rank(const mb_comm* c) { return c ? c->rank : -1; }
int mb_comm_world(const mb_comm* c) { return c ? c->world : 0; }
void* mb_comm_stream(const mb_comm* c) { return c ? (void*)c->cs : nullptr; }

static size_t al256(size_t x) { return (x + 255) / 256 * 256; }
size_t mb_comm_scratch_bytes(int world, int wire_dtype, size_t n_params, int vocab, int hidden, int capacity_rows) {
    size_t b = 0;
    if (wire_dtype == DT_BF16) b += al256(n_params * 2);
    if (vocab > 0 && capacity_rows > 0) {
        b += al256((size_t)world * vocab * sizeof(int));
        b += al256((size_t)world * capacity_rows * sizeof(int));
        b += al256((size_t)world * capacity_rows * hidden * sizeof(float));
    }
    return b ? b : 256;
}

int mb_comm_bind_scratch(mb_comm* c, void* scratch, size_t bytes, int wire_dtype, size_t n_params, int vocab, int hidden, int capacity_rows) {
    if (!c || !scratch || (wire_dtype != DT_F32 && wire_dtype != DT_BF16)) return MB_ERR_ARG;
    if (bytes < mb_comm_scratch_bytes(c->world, wire_dtype, n_params, vocab, hidden, capacity_rows)) return MB_ERR_ARG;
    c->scratch = (char*)scratch; c->scratch_bytes = bytes; c->wire = wire_dtype; c->n_params = n_params;
    c->vocab = vocab; c->H = hidden; c->cap = capacity_rows;
    size_t o = 0;
    c->off_stage = o; if (wire_dtype == DT_BF16) o += al256(n_params * 2);
    c->rows_ready = vocab > 0 && capacity_rows > 0 && hidden > 0 && hidden % 4 == 0;
    if (c->rows_ready) {
        c->off_slot = o; o += al256((size_t)c->world * vocab * sizeof(int));
        c->off_ids = o; o += al256((size_t)c->world * capacity_rows * sizeof(int));
        c->off_rows = o; o += al256((size_t)c->world * capacity_rows * hidden * sizeof(float));
        CK((int)hipMemsetAsync(c->scratch + c->off_slot, 0xFF, (size_t)c->world * vocab * sizeof(int), c->cs));     // all -1
    }
    return MB_OK;
}

int mb_comm_all_reduce(mb_comm* c, float* buf, size_t count, void* stream) {
    return comm_all_reduce(c, buf, count, (hipStream_t)stream);
}
int mb_comm_exchange_rows(mb_comm* c, float* table, const int64_t* ids, int T, void* stream) {
    return comm_exchange_rows(c, table, ids, T, (hipStream_t)stream);
}

int mb_comm_set_row_exchange(mb_comm* c, int rowwise) {
    if (!c) return MB_ERR_ARG;
    c->rowwise = rowwise != 0;
    return MB_OK;
}
int mb_comm_set_sharding(mb_comm* c, int on) {
    if (!c) return MB_ERR_ARG;
    if (on && c->gather_pending) return MB_ERR_MODE;
    // (a one-rank group has nothing to shard; MB_DP_SHARD_FORCE=1 runs the sharded code path anyway -- every collective an identity --
    //  so that its launches, event nodes and gathers can be timed on one GPU: tools/step_bench --dp 1 --shard 1)
    const char* f = getenv("MB_DP_SHARD_FORCE");
    c->shard = on != 0 && (c->world > 1 || (f && atoi(f) != 0));
    return MB_OK;
}
int mb_comm_sharding(const mb_comm* c) { return (c && c->shard) ? 1 : 0; }
int mb_comm_join(mb_comm* c, void* stream) {
    if (!c) return MB_ERR_ARG;
    return dp_step_begin(c, (hipStream_t)stream, false);
}
int mb_comm_gather_shards(mb_comm* c, void* base, int elem_bytes, void* stream) {
    if (!c || !base || (elem_bytes != 2 && elem_bytes != 4)) return MB_ERR_ARG;
    if (!c->shard) return MB_OK;
    hipStream_t st = (hipStream_t)stream;
    for (const auto& ch : c->shard_chunks) {
        const ShardSlice sl = dp_shard_slice(c, ch.first, ch.second);
        CK(comm_gather_slices(c, (char*)base, elem_bytes, ch.first, sl.per, st));
    }
    return MB_OK;
}
int mb_comm_shard_slices(const mb_comm* c, size_t* begin_end_pairs, int max_pairs) {
    if (!c || !begin_end_pairs) return 0;
    int n = 0;
    for (const auto& ch : c->shard_chunks) {
        if (n >= max_pairs) break;
        const ShardSlice sl = dp_shard_slice(c, ch.first, ch.second);
        begin_end_pairs[2 * n] = sl.mine_b; begin_end_pairs[2 * n + 1] = sl.mine_e; ++n;
    }
    return n;
}

int mb_comm_set_timing(mb_comm* c, int on) {
    if (!c) return MB_ERR_ARG;
    c->timing = on != 0;
    if (!on) c->tev_used[0] = c->tev_used[1] = false;
    return MB_OK;
}
int mb_comm_exposed_ms(mb_comm* c, float* ms) {
    if (!c || !ms) return MB_ERR_ARG;
    float total = 0.f;
    for (int k = 0; k < 2; ++k) {
        if (!c->tev_used[k]) continue;
        float t = 0.f;
        CK((int)hipEventSynchronize(c->tev[2 * k + 1]));
        CK((int)hipEventElapsedTime(&t, c->tev[2 * k], c->tev[2 * k + 1]));
        total += t;
    }
    *ms = total;
    return MB_OK;
}
int mb_comm_stats(const mb_comm* c, size_t* pieces, size_t* bytes) {
    if (!c) return MB_ERR_ARG;
    if (pieces) *pieces = c->pieces;
    if (bytes) *bytes = c->bytes_reduced;
    return MB_OK;
}

}  // extern "C"
