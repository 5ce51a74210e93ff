// Probe (measurement tooling, torch-free): what does one cross-stream hand-off cost when the event is still IN FLIGHT at the time of
// hipStreamWaitEvent?  Stream A runs a kernel (~50 us spin) and records an event; stream B waits for it, runs a tiny kernel and
// records its own event; A waits for that ... `hops` times, enqueued without any host synchronisation.  The time per round trip minus
// the two kernels' durations is what two hand-offs cost.  Variants: stream B normal / highest priority; events with / without
// hipEventDisableTiming.  Run under both runtimes of this image: as built (/opt/rocm, HIP 7.2) and with
// LD_PRELOAD=<torch>/lib/libamdhip64.so (HIP 7.0: what every PyTorch process maps) -- the data-parallel step's hand-offs
// (csrc/comm.hip) live on whichever one the host process loaded.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t _e = (x); if (_e != hipSuccess) { printf("  %s -> %s\n", #x, hipGetErrorString(_e)); return 1; } } while (0)

__global__ void spin(long long ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
}

static int run(int prio_high, unsigned evflags, int hops, long long ticks) {
    hipStream_t a, b;
    int least = 0, greatest = 0;
    CK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    CK(hipStreamCreateWithPriority(&b, hipStreamNonBlocking, prio_high ? greatest : 0));
    std::vector<hipEvent_t> ea(hops), eb(hops);
    for (auto& e : ea) CK(hipEventCreateWithFlags(&e, evflags));
    for (auto& e : eb) CK(hipEventCreateWithFlags(&e, evflags));
    // warm-up + reference: the same kernels on ONE stream
    for (int i = 0; i < 8; ++i) { spin<<<1, 64, 0, a>>>(ticks); spin<<<1, 64, 0, a>>>(100); }
    CK(hipStreamSynchronize(a));
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < hops; ++i) { spin<<<1, 64, 0, a>>>(ticks); spin<<<1, 64, 0, a>>>(100); }
    CK(hipStreamSynchronize(a));
    auto t1 = std::chrono::steady_clock::now();
    for (int i = 0; i < hops; ++i) {
        spin<<<1, 64, 0, a>>>(ticks);
        CK(hipEventRecord(ea[i], a));
        CK(hipStreamWaitEvent(b, ea[i], 0));
        spin<<<1, 64, 0, b>>>(100);
        CK(hipEventRecord(eb[i], b));
        CK(hipStreamWaitEvent(a, eb[i], 0));
    }
    CK(hipStreamSynchronize(a));
    auto t2 = std::chrono::steady_clock::now();
    const double one = std::chrono::duration<double, std::micro>(t1 - t0).count() / hops;
    const double two = std::chrono::duration<double, std::micro>(t2 - t1).count() / hops;
    printf("  comm-stream priority %-7s events %-14s: one stream %7.1f us per pair, ping-pong over two streams %8.1f us per round trip -> %7.1f us per hand-off\n",
           prio_high ? "highest" : "normal", evflags == hipEventDisableTiming ? "disable-timing" : "default", one, two, (two - one) / 2);
    for (auto e : ea) (void)hipEventDestroy(e);
    for (auto e : eb) (void)hipEventDestroy(e);
    (void)hipStreamDestroy(a); (void)hipStreamDestroy(b);
    return 0;
}

int main() {
    int rtv = 0;
    (void)hipRuntimeGetVersion(&rtv);
    printf("HIP runtime version %d\n", rtv);
    int rc = 0;
    for (int prio = 0; prio < 2; ++prio)
        for (unsigned fl : {(unsigned)hipEventDisableTiming, (unsigned)hipEventDefault}) rc |= run(prio, fl, 200, 5000 /* 50 us */);
    return rc;
}
