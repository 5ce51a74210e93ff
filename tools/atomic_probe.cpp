// atomic_probe -- throughput of uncontended, contiguous fp32 atomicAdd into an L2-sized buffer (measurement tooling, not product):
// what a split-K GEMM epilogue that adds its partial tile into an fp32 [T][768] buffer would cost.  S "splits" add to the same
// 7.4 MB buffer: block b handles a contiguous 128 x 128-float tile-shaped patch (row segments of 128 floats).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define HCK(x) do { hipError_t _e = (x); if (_e != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(_e)); exit(2); } } while (0)
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int MODE>   // 0: atomicAdd per float, 1: plain f32x4 store (the non-split epilogue), 2: packed 2 x f32 atomic (global_atomic_pk_add... n/a) -> plain RMW f32x4
__global__ void __launch_bounds__(256) k(float* buf, int ld, int tiles_n, int splits) {
    const int t = blockIdx.x / splits;
    const int tm = t / tiles_n, tn = t % tiles_n;
    const int tid = threadIdx.x;
    float* base = buf + (size_t)tm * 128 * ld + tn * 128;
    const int c = (tid & 31) * 4;
    for (int r = tid >> 5; r < 128; r += 8) {
        float* p = base + (size_t)r * ld + c;
        if (MODE == 0) {
            atomicAdd(p + 0, 1.0f); atomicAdd(p + 1, 2.0f); atomicAdd(p + 2, 3.0f); atomicAdd(p + 3, 4.0f);
        } else if (MODE == 1) {
            *(f32x4*)p = f32x4{1.f, 2.f, 3.f, 4.f};
        } else {
            f32x4 o = *(f32x4*)p; o += f32x4{1.f, 2.f, 3.f, 4.f}; *(f32x4*)p = o;
        }
    }
}

int main() {
    const int M = 2432, N = 768, tiles_m = M / 128, tiles_n = N / 128;
    float* buf; HCK(hipMalloc(&buf, (size_t)M * N * 4 * 8));
    HCK(hipMemset(buf, 0, (size_t)M * N * 4 * 8));
    hipEvent_t e0, e1; HCK(hipEventCreate(&e0)); HCK(hipEventCreate(&e1));
    for (int mode = 0; mode < 3; ++mode)
        for (int splits : {1, 2, 4}) {
            const int grid = tiles_m * tiles_n * splits, reps = 40;
            auto launch = [&](int i) {
                float* b = buf + (size_t)(i % 8) * M * N;
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, b, N, tiles_n, splits);
                else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, b, N, tiles_n, splits);
                else hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, b, N, tiles_n, splits);
            };
            for (int i = 0; i < 4; ++i) launch(i);
            HCK(hipEventRecord(e0, 0));
            for (int i = 0; i < reps; ++i) launch(i);
            HCK(hipEventRecord(e1, 0)); HCK(hipEventSynchronize(e1));
            float ms; HCK(hipEventElapsedTime(&ms, e0, e1));
            printf("%-28s splits %d (%4d blocks): %7.2f us per launch, %6.1f M lane-ops\n", mode == 0 ? "atomicAdd f32 per lane" : mode == 1 ? "plain f32x4 store" : "plain f32x4 read-modify-write",
                   splits, grid, ms * 1e3 / reps, (double)grid * 16384 / 1e6);
        }
    return 0;
}
