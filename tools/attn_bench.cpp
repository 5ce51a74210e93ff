// attn_bench -- torch-free timing of the BERT attention core through the C ABI (mb_attention_forward / _backward), with the
// per-block phase stamps of the backward when MB_ATTN_TRACE=1.  Measurement tooling (not product).
//   attn_bench [--batch B] [--seq L] [--heads nh] [--reps n] [--p dropout]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../include/magbert_hip.h"
#define HCK(x) do { hipError_t _e = (x); if (_e != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(_e)); exit(2); } } while (0)
#define MCK(x) do { int _e = (x); if (_e) { fprintf(stderr, "%s:%d magbert error %d: %s\n", __FILE__, __LINE__, _e, mb_error_string(_e)); exit(3); } } while (0)

static void* dev_rand(size_t n_bf16, uint32_t seed) {
    std::vector<uint16_t> h(n_bf16);
    uint32_t s = seed * 2654435761u + 12345u;
    for (size_t i = 0; i < n_bf16; ++i) {
        s = s * 1664525u + 1013904223u;
        const float v = ((int)((s >> 9) & 0x7FFF) - 16384) * (1.0f / 16384.f);
        uint32_t b; memcpy(&b, &v, 4);
        h[i] = (uint16_t)(b >> 16);
    }
    void* d; HCK(hipMalloc(&d, n_bf16 * 2)); HCK(hipMemcpy(d, h.data(), n_bf16 * 2, hipMemcpyHostToDevice));
    return d;
}

int main(int argc, char** argv) {
    int B = 48, L = 50, nh = 12, reps = 96; float pdrop = 0.1f;
    for (int i = 1; i + 1 < argc; i += 2) {
        std::string k = argv[i];
        if (k == "--batch") B = atoi(argv[i + 1]); else if (k == "--seq") L = atoi(argv[i + 1]); else if (k == "--heads") nh = atoi(argv[i + 1]);
        else if (k == "--reps") reps = atoi(argv[i + 1]); else if (k == "--p") pdrop = (float)atof(argv[i + 1]);
    }
    const int H = nh * 64, T = B * L, nset = 6;
    hipStream_t st; HCK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    std::vector<void*> qkv(nset), dctx(nset), ctx(nset), dqkv(nset);
    for (int s = 0; s < nset; ++s) {
        qkv[s] = dev_rand((size_t)T * 3 * H, 1 + s); dctx[s] = dev_rand((size_t)T * H, 20 + s);
        HCK(hipMalloc(&ctx[s], (size_t)T * H * 2)); HCK(hipMalloc(&dqkv[s], (size_t)T * 3 * H * 2));
    }
    std::vector<int64_t> hm((size_t)T, 1);
    for (int b = 0; b < B; ++b) for (int l = L - (b % 7); l < L; ++l) hm[(size_t)b * L + l] = 0;       // ragged tails like the padded batches
    int64_t* mask; HCK(hipMalloc(&mask, (size_t)T * 8)); HCK(hipMemcpy(mask, hm.data(), (size_t)T * 8, hipMemcpyHostToDevice));
    mb_dropkey key; mb_make_dropkey(1, 1, 17, pdrop, &key);
    hipEvent_t e0, e1; HCK(hipEventCreate(&e0)); HCK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, auto launch) {
        for (int i = 0; i < 6; ++i) launch(i);
        HCK(hipEventRecord(e0, st));
        for (int i = 0; i < reps; ++i) launch(i);
        HCK(hipEventRecord(e1, st)); HCK(hipEventSynchronize(e1));
        float ms; HCK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-28s B=%d L=%d heads=%d p=%.2f : %7.2f us/launch\n", name, B, L, nh, pdrop, ms * 1e3 / reps);
    };
    timeit("attention forward", [&](int i) { MCK(mb_attention_forward(MB_DT_BF16, qkv[i % nset], mask, ctx[i % nset], B, L, nh, &key, st)); });
    timeit("attention backward", [&](int i) { MCK(mb_attention_backward(MB_DT_BF16, qkv[i % nset], mask, dctx[i % nset], dqkv[i % nset], B, L, nh, &key, st)); });
    std::vector<unsigned long long> tr((size_t)8192 * 8);
    const int nb = mb_debug_attention_trace(tr.data(), 8192);
    if (nb > 0) {
        static const char* nm[6] = {"entry", "operands staged", "query sweep done", "key sweep done", "bias sums in LDS", "exit"};
        static const int slot[6] = {0, 1, 2, 4, 6, 5};
        unsigned long long t00 = ~0ull;
        for (int b = 0; b < nb; ++b) t00 = std::min(t00, tr[(size_t)b * 8]);
        printf("    %d blocks; us after the first block's entry (min / median / max), then per-block phase length (median)\n", nb);
        for (int k = 0; k < 6; ++k) {
            std::vector<double> a, d;
            for (int b = 0; b < nb; ++b) {
                if (!tr[(size_t)b * 8 + slot[k]]) continue;          // phase not reached (no bias gradient requested)
                a.push_back((double)(tr[(size_t)b * 8 + slot[k]] - t00) * 0.01);
                if (k) d.push_back((double)(tr[(size_t)b * 8 + slot[k]] - tr[(size_t)b * 8 + slot[k - 1]]) * 0.01);
            }
            if (a.empty()) continue;
            std::sort(a.begin(), a.end()); std::sort(d.begin(), d.end());
            printf("    %-18s %7.2f %7.2f %7.2f   %7.2f\n", nm[k], a.front(), a[a.size() / 2], a.back(), k ? d[d.size() / 2] : 0.0);
        }
    }
    return 0;
}
