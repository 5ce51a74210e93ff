// Probe (measurement tooling, torch-free): does an event record issued INSIDE a stream capture become a node that records on every
// replay, so that another stream's hipStreamWaitEvent honours it?  Answers ADVICE r4 (csrc/comm.hip event modes 2 and 3).
//   variant 0: hipEventRecord(ev, s)                         inside the capture
//   variant 1: hipEventRecordWithFlags(ev, s, External)      inside the capture
//   variant 2: hipGraphAddEventRecordNode appended after the capture
// For each: node types of the captured graph, then 20 replays on a stream other than the capture stream; each replay = one ~2 ms spin
// kernel that stores `it` at its end, [event], and on a second stream: hipStreamWaitEvent(ev) + a reader kernel.  "held" counts the
// replays in which the reader saw the value of its own iteration.
// The wait side: variant 3 = hipStreamWaitEvent(s, ev, 0) inside a capture on an event recorded outside; variant 4 = the same with
// hipEventWaitExternal; variant 5 = hipGraphAddEventWaitNode put in front of the captured root after the capture.
// Run it under BOTH runtimes a process may carry: as built (ROCm 7.2, /opt/rocm) and with
// LD_PRELOAD=<torch>/lib/libamdhip64.so (the 7.0 runtime every Python process of this image maps).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t _e = (x); if (_e != hipSuccess) { printf("  %s -> %s\n", #x, hipGetErrorString(_e)); return 1; } } while (0)

__global__ void spin_store(int* flag, int value, long long ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
    *flag = value;
}
__global__ void read_flag(const int* flag, int* out, int slot) { out[slot] = *(volatile const int*)flag; }
__global__ void store_param(int* flag, const int* src) { *flag = *src; }

static const char* tname(hipGraphNodeType t) {
    switch (t) {
        case hipGraphNodeTypeKernel: return "kernel";
        case hipGraphNodeTypeEventRecord: return "event-record";
        case hipGraphNodeTypeWaitEvent: return "wait-event";
        case hipGraphNodeTypeEmpty: return "empty";
        default: return "other";
    }
}
static int show_nodes(hipGraph_t g) {
    size_t n = 0;
    CK(hipGraphGetNodes(g, nullptr, &n));
    std::vector<hipGraphNode_t> nodes(n);
    if (n) CK(hipGraphGetNodes(g, nodes.data(), &n));
    printf("  nodes:");
    for (auto nd : nodes) { hipGraphNodeType t; CK(hipGraphNodeGetType(nd, &t)); printf(" %s", tname(t)); }
    printf("\n");
    return 0;
}

static int record_variant(int variant) {
    printf("record variant %d\n", variant);
    hipStream_t cap, run, side;
    CK(hipStreamCreateWithFlags(&cap, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&run, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
    hipEvent_t ev;
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    int *flag, *out, *itv;
    CK(hipMalloc(&flag, 4)); CK(hipMalloc(&out, 4 * 64)); CK(hipMalloc(&itv, 4));
    CK(hipMemset(flag, 0, 4)); CK(hipMemset(out, 0xFF, 4 * 64));
    const long long ticks = 200000;      // wall_clock64 runs at 100 MHz: 2 ms
    hipGraph_t g = nullptr; hipGraphExec_t ex = nullptr;
    CK(hipStreamBeginCapture(cap, hipStreamCaptureModeRelaxed));
    spin_store<<<1, 1, 0, cap>>>(flag, -7, ticks);            // (value replaced below: the flag takes the iteration from itv)
    store_param<<<1, 1, 0, cap>>>(flag, itv);
    if (variant == 0) CK(hipEventRecord(ev, cap));
    if (variant == 1) {
        hipError_t er = hipEventRecordWithFlags(ev, cap, hipEventRecordExternal);
        if (er != hipSuccess) { printf("  hipEventRecordWithFlags(External) inside the capture -> %s\n", hipGetErrorString(er)); (void)hipGetLastError(); }
    }
    CK(hipStreamEndCapture(cap, &g));
    if (variant == 2) {
        size_t n = 0; CK(hipGraphGetNodes(g, nullptr, &n));
        std::vector<hipGraphNode_t> nodes(n); CK(hipGraphGetNodes(g, nodes.data(), &n));
        size_t nl = 0; CK(hipGraphGetRootNodes(g, nullptr, &nl));
        // the last captured node = the one nothing depends on: for a linear capture, find it via edges
        hipGraphNode_t last = nullptr;
        for (auto nd : nodes) { size_t nd_out = 0; CK(hipGraphNodeGetDependentNodes(nd, nullptr, &nd_out)); if (nd_out == 0) last = nd; }
        hipGraphNode_t rec;
        CK(hipGraphAddEventRecordNode(&rec, g, &last, 1, ev));
    }
    if (show_nodes(g)) return 1;
    CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
    int held = 0, err = 0;
    for (int it = 1; it <= 20; ++it) {
        CK(hipMemcpyAsync(itv, &it, 4, hipMemcpyHostToDevice, run));      // (pageable: returns after the copy was staged)
        CK(hipGraphLaunch(ex, run));
        hipError_t w = hipStreamWaitEvent(side, ev, 0);
        if (w != hipSuccess) { if (!err) printf("  hipStreamWaitEvent after replay -> %s\n", hipGetErrorString(w)); ++err; (void)hipGetLastError(); }
        read_flag<<<1, 1, 0, side>>>(flag, out, it);
        CK(hipDeviceSynchronize());
    }
    int h[64];
    CK(hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost));
    for (int it = 1; it <= 20; ++it) held += h[it] == it;
    printf("  dependency held in %d of 20 replays (wait errors: %d)\n", held, err);
    return 0;
}

static int wait_variant(int variant) {
    printf("wait variant %d\n", variant);
    hipStream_t cap, run, side;
    CK(hipStreamCreateWithFlags(&cap, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&run, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
    hipEvent_t ev;
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    int *flag, *out;
    CK(hipMalloc(&flag, 4)); CK(hipMalloc(&out, 4 * 64));
    CK(hipMemset(flag, 0, 4)); CK(hipMemset(out, 0xFF, 4 * 64));
    hipGraph_t g = nullptr; hipGraphExec_t ex = nullptr;
    CK(hipEventRecord(ev, side));          // a first record outside any capture
    CK(hipStreamBeginCapture(cap, hipStreamCaptureModeRelaxed));
    hipError_t w = variant == 3 ? hipStreamWaitEvent(cap, ev, 0) : variant == 4 ? hipStreamWaitEvent(cap, ev, hipEventWaitExternal) : hipSuccess;
    if (w != hipSuccess) { printf("  wait inside the capture -> %s\n", hipGetErrorString(w)); (void)hipGetLastError(); }
    read_flag<<<1, 1, 0, cap>>>(flag, out, 0);
    hipError_t e2 = hipStreamEndCapture(cap, &g);
    if (e2 != hipSuccess) { printf("  hipStreamEndCapture -> %s\n", hipGetErrorString(e2)); return 0; }
    if (variant == 5) {
        size_t nr = 0; CK(hipGraphGetRootNodes(g, nullptr, &nr));
        std::vector<hipGraphNode_t> roots(nr); CK(hipGraphGetRootNodes(g, roots.data(), &nr));
        hipGraphNode_t wn;
        CK(hipGraphAddEventWaitNode(&wn, g, nullptr, 0, ev));
        for (auto r : roots) CK(hipGraphAddDependencies(g, &wn, &r, 1));
    }
    if (show_nodes(g)) return 1;
    CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
    // the reader of the graph stores out[0]; copy it to out[it] afterwards on the same stream
    int held = 0;
    for (int it = 1; it <= 20; ++it) {
        spin_store<<<1, 1, 0, side>>>(flag, it, 200000);
        CK(hipEventRecord(ev, side));
        CK(hipGraphLaunch(ex, run));
        CK(hipMemcpyAsync(out + it, out, 4, hipMemcpyDeviceToDevice, run));
        CK(hipDeviceSynchronize());
    }
    int h[64];
    CK(hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost));
    for (int it = 1; it <= 20; ++it) held += h[it] == it;
    printf("  dependency held in %d of 20 replays\n", held);
    return 0;
}

int main() {
    int rc = 0, rtv = 0;
    (void)hipRuntimeGetVersion(&rtv);
    printf("HIP runtime version %d\n", rtv);
    for (int v = 0; v < 3; ++v) rc |= record_variant(v);
    for (int v = 3; v < 6; ++v) rc |= wait_variant(v);
    return rc;
}
