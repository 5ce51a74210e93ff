// launch_floor -- what one kernel launch costs on the stream before it does any work (measurement tooling, not product).
// Times back-to-back launches (stream and replayed hipGraph) of kernels that do nothing / allocate LDS / write an output of the
// size of a GEMM epilogue, so that per-kernel fixed cost can be separated from main-loop time in profiles/.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define HCK(x) do { hipError_t _e = (x); if (_e != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(_e)); exit(2); } } while (0)

__global__ void __launch_bounds__(256) k_empty(int* p) { if (p && threadIdx.x == 9999) *p = 1; }
template <int KB>
__global__ void __launch_bounds__(256) k_lds(int* p) {
    __shared__ char smem[KB * 1024];
    smem[threadIdx.x] = (char)threadIdx.x;
    __syncthreads();
    if (p && smem[(threadIdx.x + 1) & 255] == 77 && threadIdx.x == 9999) *p = 1;
}
// every thread writes `n16` 16-byte vectors, a block covers a contiguous piece (the shape of a row-major epilogue pass)
__global__ void __launch_bounds__(256) k_store(float4* out, int n16) {
    float4 v = {1.f, 2.f, 3.f, 4.f};
    float4* o = out + (size_t)blockIdx.x * 256 * n16 + threadIdx.x;
    for (int i = 0; i < n16; ++i) o[(size_t)i * 256] = v;
}
__global__ void __launch_bounds__(256) k_load_store(const float4* in, float4* out, int n16) {
    const float4* a = in + (size_t)blockIdx.x * 256 * n16 + threadIdx.x;
    float4* o = out + (size_t)blockIdx.x * 256 * n16 + threadIdx.x;
    for (int i = 0; i < n16; ++i) { float4 v = a[(size_t)i * 256]; v.x += 1.f; o[(size_t)i * 256] = v; }
}

template <class F> static void timeit(const char* name, hipStream_t st, int reps, F launch) {
    hipEvent_t e0, e1; HCK(hipEventCreate(&e0)); HCK(hipEventCreate(&e1));
    for (int i = 0; i < 20; ++i) launch(i);
    HCK(hipStreamSynchronize(st));
    HCK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i) launch(i);
    HCK(hipEventRecord(e1, st));
    HCK(hipEventSynchronize(e1));
    float ms; HCK(hipEventElapsedTime(&ms, e0, e1));
    // the same launches as one replayed graph
    hipGraph_t g; hipGraphExec_t ge;
    HCK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < reps; ++i) launch(i);
    HCK(hipStreamEndCapture(st, &g));
    HCK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    HCK(hipGraphLaunch(ge, st)); HCK(hipStreamSynchronize(st));
    HCK(hipEventRecord(e0, st));
    HCK(hipGraphLaunch(ge, st));
    HCK(hipEventRecord(e1, st));
    HCK(hipEventSynchronize(e1));
    float msg; HCK(hipEventElapsedTime(&msg, e0, e1));
    printf("%-58s %7.2f us/launch (stream) %7.2f us/launch (graph)\n", name, ms * 1e3 / reps, msg * 1e3 / reps);
    HCK(hipGraphExecDestroy(ge)); HCK(hipGraphDestroy(g));
}

int main() {
    hipStream_t st; HCK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const int reps = 200, nset = 6;
    const size_t bytes = (size_t)456 * 256 * 16 * 16;     // 456 blocks x 256 threads x 16 x 16 B = 29.9 MB
    std::vector<float4*> in(nset), out(nset);
    for (int s = 0; s < nset; ++s) { HCK(hipMalloc(&in[s], bytes)); HCK(hipMalloc(&out[s], bytes)); HCK(hipMemset(in[s], 0, bytes)); }
    timeit("empty, 1 block", st, reps, [&](int) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(256), 0, st, (int*)nullptr); });
    timeit("empty, 456 blocks", st, reps, [&](int) { hipLaunchKernelGGL(k_empty, dim3(456), dim3(256), 0, st, (int*)nullptr); });
    timeit("empty, 4096 blocks", st, reps, [&](int) { hipLaunchKernelGGL(k_empty, dim3(4096), dim3(256), 0, st, (int*)nullptr); });
    timeit("64 KB LDS touched, 456 blocks", st, reps, [&](int) { hipLaunchKernelGGL(k_lds<64>, dim3(456), dim3(256), 0, st, (int*)nullptr); });
    timeit("144 KB LDS touched, 240 blocks", st, reps, [&](int) { hipLaunchKernelGGL(k_lds<144>, dim3(240), dim3(256), 0, st, (int*)nullptr); });
    timeit("store 3.7 MB (456 blocks x 2 x 16 B/thread)", st, reps, [&](int i) { hipLaunchKernelGGL(k_store, dim3(456), dim3(256), 0, st, out[i % nset], 2); });
    timeit("store 11 MB (456 x 6)", st, reps, [&](int i) { hipLaunchKernelGGL(k_store, dim3(456), dim3(256), 0, st, out[i % nset], 6); });
    timeit("store 29.9 MB (456 x 16)", st, reps, [&](int i) { hipLaunchKernelGGL(k_store, dim3(456), dim3(256), 0, st, out[i % nset], 16); });
    timeit("load+store 3.7 MB each", st, reps, [&](int i) { hipLaunchKernelGGL(k_load_store, dim3(456), dim3(256), 0, st, in[i % nset], out[i % nset], 2); });
    timeit("load+store 29.9 MB each", st, reps, [&](int i) { hipLaunchKernelGGL(k_load_store, dim3(456), dim3(256), 0, st, in[i % nset], out[i % nset], 16); });
    timeit("load+store 29.9 MB each, 1824 blocks x 4", st, reps, [&](int i) { hipLaunchKernelGGL(k_load_store, dim3(1824), dim3(256), 0, st, in[i % nset], out[i % nset], 4); });
    // what ONE fork / join costs a replayed graph: 200 small kernels in a chain vs the same chain with kernels 100..119 on a second
    // stream next to kernels 120..139 (event fork + event join captured as graph dependencies)
    {
        hipStream_t s2; HCK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
        hipEvent_t f, j, e0, e1; HCK(hipEventCreateWithFlags(&f, hipEventDisableTiming)); HCK(hipEventCreateWithFlags(&j, hipEventDisableTiming));
        HCK(hipEventCreate(&e0)); HCK(hipEventCreate(&e1));
        for (int branches = 0; branches <= 12; branches = branches ? branches * 12 : 1) {
            hipGraph_t g; hipGraphExec_t ge;
            HCK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            const int per = branches ? 200 / branches : 200;
            for (int i = 0; i < 200; ++i) {
                const bool fork_here = branches && (i % per) == per / 2;
                if (fork_here) {
                    HCK(hipEventRecord(f, st)); HCK(hipStreamWaitEvent(s2, f, 0));
                    for (int k = 0; k < 4; ++k) hipLaunchKernelGGL(k_store, dim3(456), dim3(256), 0, s2, out[1], 2);
                    HCK(hipEventRecord(j, s2));
                }
                hipLaunchKernelGGL(k_store, dim3(456), dim3(256), 0, st, out[0], 2);
                if (branches && (i % per) == per / 2 + 4) HCK(hipStreamWaitEvent(st, j, 0));
            }
            HCK(hipStreamEndCapture(st, &g));
            HCK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            for (int w = 0; w < 3; ++w) HCK(hipGraphLaunch(ge, st));
            HCK(hipStreamSynchronize(st));
            HCK(hipEventRecord(e0, st));
            for (int r = 0; r < 10; ++r) HCK(hipGraphLaunch(ge, st));
            HCK(hipEventRecord(e1, st)); HCK(hipEventSynchronize(e1));
            float ms; HCK(hipEventElapsedTime(&ms, e0, e1));
            printf("graph of 200 chained 3.7-MB store kernels, %2d fork/join(s) with 4 extra kernels each on a second stream: %8.1f us per replay\n",
                   branches, ms * 100.0);
            HCK(hipGraphExecDestroy(ge)); HCK(hipGraphDestroy(g));
        }
    }
    return 0;
}
