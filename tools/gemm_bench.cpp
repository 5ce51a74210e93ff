// gemm_bench -- torch-free timing of the encoder's GEMM launches through the C ABI (mb_gemm / mb_gemm_grouped_wgrad).
// Measurement tooling (not product).  Every case runs `reps` launches over `nset` rotating operand sets (so that operands
// come from HBM / Infinity Cache like inside a training step, not from a warm L2) between two HIP events.
//
//   gemm_bench [--T tokens] [--reps n] [--nset n] [--only substring] [--trace 1]
//   --trace 1 (with MB_GEMM_TRACE=1): after timing a case, one more launch whose per-block phase stamps are summarised
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <algorithm>
#include "../include/magbert_hip.h"

#define HCK(x) do { hipError_t _e = (x); if (_e != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(_e)); exit(2); } } while (0)
#define MCK(x) do { int _e = (x); if (_e) { fprintf(stderr, "%s:%d magbert error %d: %s\n", __FILE__, __LINE__, _e, mb_error_string(_e)); exit(3); } } while (0)

static void* dev_rand(size_t n_bf16, uint32_t seed) {
    std::vector<uint16_t> h(n_bf16);
    uint32_t s = seed * 2654435761u + 12345u;
    for (size_t i = 0; i < n_bf16; ++i) {
        s = s * 1664525u + 1013904223u;
        const float v = ((int)((s >> 9) & 0x7FFF) - 16384) * (0.05f / 16384.f);        // uniform in [-0.05, 0.05)
        uint32_t b; memcpy(&b, &v, 4);
        h[i] = (uint16_t)(b >> 16);
    }
    void* d; HCK(hipMalloc(&d, n_bf16 * 2)); HCK(hipMemcpy(d, h.data(), n_bf16 * 2, hipMemcpyHostToDevice));
    return d;
}

int main(int argc, char** argv) {
    int T = 2400, reps = 48, nset = 6, trace = 0, looptrace = 0, tile = 0, wtile = 128;
    std::string only;
    for (int i = 1; i + 1 < argc; i += 2) {
        std::string k = argv[i];
        if (k == "--T") T = atoi(argv[i + 1]); else if (k == "--reps") reps = atoi(argv[i + 1]);
        else if (k == "--nset") nset = atoi(argv[i + 1]); else if (k == "--only") only = argv[i + 1];
        else if (k == "--trace") trace = atoi(argv[i + 1]);
        else if (k == "--tile") tile = atoi(argv[i + 1]);          // mb_gemm tile code for every case (0 = auto, 64 | 128 | 256)
        else if (k == "--wtile") wtile = atoi(argv[i + 1]);        // grouped weight-gradient tile (64 | 128 | 256 = 256 x 128 ping-pong)
        else if (k == "--looptrace") { looptrace = atoi(argv[i + 1]); trace = trace || looptrace; }   // library built with -DMB_GEMM_LOOPTRACE
    }
    const int H = 768, I = 3072;
    const int Tp = (T + 63) / 64 * 64;
    hipStream_t st; HCK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    std::vector<void*> xh(nset), xi(nset), x3(nset), oi(nset), oi2(nset), o3(nset);
    for (int s = 0; s < nset; ++s) {
        xh[s] = dev_rand((size_t)Tp * H, 10 + s); xi[s] = dev_rand((size_t)Tp * I, 20 + s); x3[s] = dev_rand((size_t)Tp * 3 * H, 30 + s);
        HCK(hipMalloc(&oi[s], (size_t)Tp * I * 2)); HCK(hipMalloc(&oi2[s], (size_t)Tp * I * 2)); HCK(hipMalloc(&o3[s], (size_t)Tp * 3 * H * 2));
    }
    void *wqkv = dev_rand((size_t)3 * H * H, 1), *wo = dev_rand((size_t)H * H, 2), *w1 = dev_rand((size_t)I * H, 3), *w2 = dev_rand((size_t)H * I, 4);
    float *bias, *colsum, *gW[4];
    HCK(hipMalloc(&bias, (size_t)I * 4)); HCK(hipMemset(bias, 0, (size_t)I * 4));
    HCK(hipMalloc(&colsum, (size_t)I * 4)); HCK(hipMemset(colsum, 0, (size_t)I * 4));
    const int gM[4] = {H, I, H, 3 * H}, gN[4] = {I, H, H, H};
    for (int g = 0; g < 4; ++g) { HCK(hipMalloc(&gW[g], (size_t)gM[g] * gN[g] * 4)); HCK(hipMemset(gW[g], 0, (size_t)gM[g] * gN[g] * 4)); }
    mb_dropkey key; mb_make_dropkey(1, 1, 17, 0.1f, &key);
    enum { NT = 0, NN = 1 };
    struct Case { const char* name; int layout, epi, M, N, K; int a, b, r, c; };     // operand selectors: 0 xh, 1 xi, 2 x3 ; weights 0 qkv 1 o 2 w1 3 w2
    const Case cases[] = {
        {"fwd qkv   [T,768]x[2304,768]^T +bias", NT, 0, T, 3 * H, H, 0, 0, -1, 2},
        {"fwd out   [T,768]x[768,768]^T +bias+drop+res", NT, 2, T, H, H, 0, 1, 0, 0},
        {"fwd ffn1  [T,768]x[3072,768]^T +bias+gelu", NT, 1, T, I, H, 0, 2, -1, 1},
        {"fwd ffn2  [T,3072]x[768,3072]^T +bias+drop+res", NT, 2, T, H, I, 1, 3, 0, 0},
        {"dgrad ffn2 [T,768]x[768,3072] *gelu' +colsum", NN, 4, T, I, H, 0, 3, 1, 1},
        {"dgrad ffn1 [T,3072]x[3072,768] +res", NN, 3, T, H, I, 1, 2, 0, 0},
        {"dgrad out  [T,768]x[768,768]", NN, 3, T, H, H, 0, 1, -1, 0},
        {"dgrad qkv  [T,2304]x[2304,768] +res", NN, 3, T, H, 3 * H, 2, 0, 0, 0},
        // --only probe: half the k range of the three N = 768, K >= 2304 launches -- what ONE block of a two-way split-K would run
        // (with --tile 128: 114 blocks, one per CU, as 228 such blocks would sit on 228 CUs)
        {"probe half-K ffn2  [T,1536]x[768,1536]^T +bias+drop+res", NT, 2, T, H, I / 2, 1, 3, 0, 0},
        {"probe half-K dffn1 [T,1536]x[1536,768] +res", NN, 3, T, H, I / 2, 1, 2, 0, 0},
        {"probe half-K dqkv  [T,1152]x[1152,768] +res", NN, 3, T, H, 3 * H / 2, 2, 0, 0, 0},
    };
    void* W[4] = {wqkv, wo, w1, w2};
    const int ldw[4] = {H, H, H, I};
    hipEvent_t e0, e1; HCK(hipEventCreate(&e0)); HCK(hipEventCreate(&e1));
    double tot_us = 0, tot_fl = 0;
    auto sel = [&](int which, int s) -> void* { return which == 0 ? xh[s] : which == 1 ? xi[s] : x3[s]; };
    auto ld = [&](int which) { return which == 0 ? H : which == 1 ? I : 3 * H; };
    auto trace_summary = [&]() {
        // phase picture of ONE launch in steady state (the queue is kept busy by the launches in front of it)
        const size_t S = looptrace ? 128 : 8;          // u64 per block
        std::vector<unsigned long long> tr((size_t)8192 * 8);
        const int nb = mb_debug_gemm_trace(tr.data(), (int)(8192 * 8 / S));
        std::vector<double> ph[5];
        unsigned long long t00 = ~0ull;
        for (int b = 0; b < nb; ++b) if (tr[(size_t)b * S]) t00 = std::min(t00, tr[(size_t)b * S]);
        for (int b = 0; b < nb; ++b) {
            if (!tr[(size_t)b * S]) continue;
            for (int k = 0; k < 5; ++k) ph[k].push_back((double)(tr[(size_t)b * S + k] - t00) * 0.01);
        }
        if (looptrace == 2) {
            // gemm_pp.hip: waves 0 (group 0) and 4 (group 1), 10 k-stages x 6 stamps each
            static const char* pn[6] = {"LOAD: issue reads", "LOAD: reads return (+g1: stage landed)", "barrier 1", "COMP: 32 MFMAs + DMA",
                                        "COMP: stage landed (g0)", "barrier 2"};
            constexpr int NP = 6, NI = 10;
            for (int grp = 0; grp < 2; ++grp) {
                std::vector<double> d[NP + 1];
                for (int b = 0; b < nb; ++b) {
                    if (!tr[(size_t)b * S]) continue;
                    const unsigned long long* lt = &tr[(size_t)b * S + 8 + grp * NP * NI];
                    for (int t = 0; t + 1 < NI; ++t) {
                        if (!lt[(t + 1) * NP]) break;
                        auto df = [&](int i1, int i0) { return (double)(uint32_t)((uint32_t)lt[i1] - (uint32_t)lt[i0]); };
                        for (int k = 0; k < NP - 1; ++k) d[k].push_back(df(t * NP + k + 1, t * NP + k));
                        d[NP - 1].push_back(df((t + 1) * NP, t * NP + NP - 1));
                        d[NP].push_back(df((t + 1) * NP, t * NP));
                    }
                }
                printf("    group %d: k-stage of wave %d, shader clocks (p10 / median / p90 / mean over %d samples)\n", grp, grp * 4, (int)d[0].size());
                for (int k = 0; k <= NP; ++k) {
                    if (d[k].empty()) continue;
                    std::sort(d[k].begin(), d[k].end());
                    double m = 0; for (double x : d[k]) m += x;
                    printf("    %-40s %7.0f %7.0f %7.0f %7.0f\n", k < NP ? pn[k] : "whole stage", d[k][d[k].size() / 10], d[k][d[k].size() / 2], d[k][d[k].size() * 9 / 10], m / d[k].size());
                }
            }
        } else if (looptrace) {
            // shader-clock stamps of wave 0, iterations 4 .. 22 of every block: where an iteration of the k loop goes
            static const char* pn[5] = {"wait for the stage (vmcnt)", "barrier", "DMA issue", "fragment reads + MFMA issue", "whole iteration"};
            std::vector<double> d[5];
            for (int b = 0; b < nb; ++b) {
                if (!tr[(size_t)b * S]) continue;
                const unsigned long long* lt = &tr[(size_t)b * S + 8];
                for (int t = 4; t < 22; ++t) {
                    if (!lt[(t + 1) * 5]) break;
                    auto df = [&](int i1, int i0) { return (double)(uint32_t)((uint32_t)lt[i1] - (uint32_t)lt[i0]); };
                    d[0].push_back(df(t * 5 + 1, t * 5)); d[1].push_back(df(t * 5 + 2, t * 5 + 1)); d[2].push_back(df(t * 5 + 3, t * 5 + 2));
                    d[3].push_back(df(t * 5 + 4, t * 5 + 3)); d[4].push_back(df((t + 1) * 5, t * 5));
                }
            }
            printf("    k-loop iteration of wave 0, shader clocks (p10 / median / p90 / mean over %d samples)\n", (int)d[0].size());
            for (int k = 0; k < 5; ++k) {
                if (d[k].empty()) continue;
                std::sort(d[k].begin(), d[k].end());
                double m = 0; for (double x : d[k]) m += x;
                printf("    %-28s %7.0f %7.0f %7.0f %7.0f\n", pn[k], d[k][d[k].size() / 10], d[k][d[k].size() / 2], d[k][d[k].size() * 9 / 10], m / d[k].size());
            }
        }
        static const char* nm[5] = {"entry", "stage 0 landed", "k loop done", "epilogue issued", "stores done"};
        printf("    %d blocks with a tile of %d launched; us after the first block's entry  (min / median / max)\n", (int)ph[0].size(), nb);
        for (int k = 0; k < 5; ++k) {
            if (ph[k].empty()) continue;
            std::sort(ph[k].begin(), ph[k].end());
            printf("    %-16s %7.2f %7.2f %7.2f\n", nm[k], ph[k].front(), ph[k][ph[k].size() / 2], ph[k].back());
        }
    };
    for (const Case& c : cases) {
        if (!only.empty() && !strstr(c.name, only.c_str())) continue;
        if (only.empty() && strstr(c.name, "probe")) continue;          // (the probes only on request)
        auto launch = [&](int i) {
            const int s = i % nset;
            void* out = c.c == 0 ? o3[s] : c.c == 1 ? oi[s] : o3[s];
            // NN: B is the weight as stored [K][N] (ldb = N); NT: B [N][K] (ldb = K)
            const int ldb = c.layout == NN ? c.N : c.K;
            (void)ldw;
            MCK(mb_gemm(MB_DT_BF16, c.layout, c.epi, c.M, c.N, c.K, sel(c.a, s), ld(c.a), W[c.b], ldb, out, c.N, oi2[s],
                        c.epi == 4 ? colsum : nullptr, bias, c.r >= 0 ? sel(c.r, (s + 1) % nset) : nullptr, c.r >= 0 ? ld(c.r) : 0, 1.0f, &key, 1, tile, st));
        };
        for (int i = 0; i < 4; ++i) launch(i);
        HCK(hipEventRecord(e0, st));
        for (int i = 0; i < reps; ++i) launch(i);
        HCK(hipEventRecord(e1, st)); HCK(hipEventSynchronize(e1));
        float ms; HCK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / reps, fl = 2.0 * c.M * c.N * c.K;
        printf("%-52s %8.2f us %8.1f TF/s\n", c.name, us, fl / us * 1e-6);
        tot_us += us; tot_fl += fl;
        if (trace) { for (int i = 0; i < 8; ++i) launch(i); trace_summary(); }
    }
    if (only.empty() || strstr("wgrad", only.c_str())) {
        auto launch = [&](int i) {
            const int s = i % nset;
            const void* dY[4] = {xh[s], xi[s], xh[(s + 1) % nset], x3[s]};
            const void* X[4] = {xi[(s + 2) % nset], xh[(s + 2) % nset], xh[(s + 3) % nset], xh[(s + 4) % nset]};
            const int ldy[4] = {H, I, H, 3 * H}, ldx[4] = {I, H, H, H};
            MCK(mb_gemm_grouped_wgrad(MB_DT_BF16, 4, gM, gN, Tp, dY, ldy, X, ldx, gW, gN, wtile, st));
        };
        for (int i = 0; i < 4; ++i) launch(i);
        HCK(hipEventRecord(e0, st));
        for (int i = 0; i < reps; ++i) launch(i);
        HCK(hipEventRecord(e1, st)); HCK(hipEventSynchronize(e1));
        float ms; HCK(hipEventElapsedTime(&ms, e0, e1));
        double fl = 0; for (int g = 0; g < 4; ++g) fl += 2.0 * gM[g] * gN[g] * Tp;
        const double us = ms * 1e3 / reps;
        printf("%-52s %8.2f us %8.1f TF/s\n", "wgrad x4 grouped [768x3072|3072x768|768x768|2304x768]", us, fl / us * 1e-6);
        tot_us += us; tot_fl += fl;
        if (trace) { for (int i = 0; i < 8; ++i) launch(i); trace_summary(); }
    }
    printf("per-layer GEMM time %.1f us, aggregate %.1f TF/s (T=%d, bf16, %d rotating operand sets)\n", tot_us, tot_fl / tot_us * 1e-6, T, nset);
    return 0;
}
