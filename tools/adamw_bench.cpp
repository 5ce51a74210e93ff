// adamw_bench -- torch-free timing of the optimizer launches of one step (mb_adamw_step over the flat buffers of the bench model:
// 110.85 M parameters in two groups, bf16 shadow over the GEMM weights).  Measurement tooling, not product.
//   adamw_bench [--reps n] [--zero 0|1]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../include/magbert_hip.h"
#define HCK(x) do { hipError_t _e = (x); if (_e != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(_e)); exit(2); } } while (0)
#define MCK(x) do { int _e = (x); if (_e) { fprintf(stderr, "%s:%d magbert error %d\n", __FILE__, __LINE__, _e); exit(3); } } while (0)
int main(int argc, char** argv) {
    int reps = 20, zero = 1;
    for (int i = 1; i + 1 < argc; i += 2) { std::string k = argv[i]; if (k == "--reps") reps = atoi(argv[i + 1]); else if (k == "--zero") zero = atoi(argv[i + 1]); }
    const size_t n = 110853184, nd = 110733312, she = 85524480;     // bench model: total, decay group, shadow range
    float *p, *g, *m, *v; void* sh;
    HCK(hipMalloc(&p, n * 4)); HCK(hipMalloc(&g, n * 4)); HCK(hipMalloc(&m, n * 4)); HCK(hipMalloc(&v, n * 4)); HCK(hipMalloc(&sh, n * 2));
    HCK(hipMemset(p, 0, n * 4)); HCK(hipMemset(g, 0, n * 4)); HCK(hipMemset(m, 0, n * 4)); HCK(hipMemset(v, 0, n * 4));
    hipStream_t st; HCK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    auto run = [&](int t) {
        MCK(mb_adamw_step(p, g, m, v, sh, nd, nd, 0, she, 1e-5f, 0.9f, 0.999f, 1e-6f, 0.01f, t, 1, 1.0f, zero, st));
        MCK(mb_adamw_step(p + nd, g + nd, m + nd, v + nd, nullptr, n - nd, 0, 0, 0, 1e-5f, 0.9f, 0.999f, 1e-6f, 0.f, t, 1, 1.0f, zero, st));
    };
    for (int i = 0; i < 3; ++i) run(i + 1);
    hipEvent_t e0, e1; HCK(hipEventCreate(&e0)); HCK(hipEventCreate(&e1));
    HCK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i) run(i + 4);
    HCK(hipEventRecord(e1, st)); HCK(hipEventSynchronize(e1));
    float ms; HCK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    const double bytes = (double)n * (zero ? 32 : 28) + (double)she * 2;
    printf("adamw update of %.2f M parameters (zero_grad=%d): %.1f us, %.2f TB/s over %.0f MB\n", n * 1e-6, zero, ms * 1e3, bytes / ms * 1e-9, bytes * 1e-6);
    return 0;
}
