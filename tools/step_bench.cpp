// step_bench -- torch-free driver of the MAG-BERT step engine through the C ABI (include/magbert_hip.h).
//
// Measurement tooling (not product): runs the same optimizer step bench.py times -- forward + fused MSE + backward +
// HF-AdamW on one synthetic minibatch in prepare_bert_input's layout (/root/reference/multimodal_driver.py:143-180) -- from a
// plain C++ process, so a GPU-box visit costs seconds instead of a Python/torch start-up, and rocprofv3 can wrap it.
//
//   step_bench [--model bert|xlnet] [--steps K] [--warmup W] [--batch B] [--seq L] [--dtype bf16|fp32] [--visual V] [--layers N]
//              [--graph 0|1|2] [--h2d 0|1|2] [--nbatch n] [--dp 0|1] [--wire fp32|bf16] [--sparse 0|1] [--timing 0|1] [--shard 0|1]
//   --dp 1:  the data-parallel step, mb_bert_train_step_dp, with a ONE-rank RCCL communicator created here through the C ABI
//            (mb_comm_unique_id / mb_comm_create_rccl): the N > 1 code path -- graph chain, comm stream, events, ncclAllReduce /
//            ncclAllGather calls, the row-wise word-embedding exchange -- without a second GPU.  Needs --graph 1|2.
//   --graph: 0 forward/backward/AdamW calls, 1 mb_bert_train_step hipGraph replay, 2 mb_bert_train_step stream launches
//   --h2d:   0 batch resident in HBM, 1 hipMemcpyAsync per step, 2 batch read in place from pinned host memory by the prologue
//
// Prints one line per run: ms/step (HIP events around the K timed steps), host enqueue ms/step, samples/s, final loss.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../include/magbert_hip.h"

#define HCK(x) do { hipError_t _e = (x); if (_e != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(_e)); exit(2); } } while (0)
#define MCK(x) do { int _e = (x); if (_e) { fprintf(stderr, "%s:%d magbert error %d: %s\n", __FILE__, __LINE__, _e, mb_error_string(_e)); exit(3); } } while (0)

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static inline uint64_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }
static inline float urand() { return (float)((rnd() >> 40) * (1.0 / 16777216.0)); }
static inline float nrand() { float u = urand() + 1e-7f, v = urand(); return sqrtf(-2.f * logf(u)) * cosf(6.2831853f * v); }

struct Batch { int64_t *ids, *seg, *mask; float *vis, *aco, *lab; };

int main(int argc, char** argv) {
    int steps = 30, warmup = 5, B = 48, L = 50, V = 47, A = 74, layers = 12, graph = 0, h2d = 0, nbatch = 4, dtype = MB_DT_BF16;
    int dp = 0, wire = MB_DT_F32, sparse = 1, timing = 0, shard = 0, xl = 0;
    for (int i = 1; i + 1 < argc; i += 2) {
        std::string k = argv[i]; const char* v = argv[i + 1];
        if (k == "--steps") steps = atoi(v); else if (k == "--warmup") warmup = atoi(v); else if (k == "--batch") B = atoi(v);
        else if (k == "--seq") L = atoi(v); else if (k == "--visual") V = atoi(v); else if (k == "--layers") layers = atoi(v);
        else if (k == "--graph") graph = atoi(v); else if (k == "--h2d") h2d = atoi(v); else if (k == "--nbatch") nbatch = atoi(v);
        else if (k == "--dtype") dtype = strcmp(v, "fp32") == 0 ? MB_DT_F32 : MB_DT_BF16;
        else if (k == "--dp") dp = atoi(v); else if (k == "--sparse") sparse = atoi(v); else if (k == "--timing") timing = atoi(v);
        else if (k == "--shard") shard = atoi(v);
        else if (k == "--model") xl = strcmp(v, "xlnet") == 0;       // MAG-XLNet (BASELINE.json configs[3]): the single-call step only (--graph 1|2)
        else if (k == "--wire") wire = strcmp(v, "bf16") == 0 ? MB_DT_BF16 : MB_DT_F32;
        else { fprintf(stderr, "unknown option %s\n", k.c_str()); return 1; }
    }
    mb_bert_config c = {};
    c.vocab_size = 30522; c.hidden_size = 768; c.num_layers = layers; c.num_heads = 12; c.intermediate_size = 3072;
    c.max_position = 512; c.type_vocab = 2; c.num_labels = 1; c.visual_dim = V; c.acoustic_dim = A; c.pad_token_id = 0;
    c.layer_norm_eps = 1e-12f; c.mag_layer_norm_eps = 1e-5f; c.beta_shift = 1.0f;
    c.hidden_dropout = 0.1f; c.attn_dropout = 0.1f; c.mag_dropout = 0.5f; c.dtype = dtype; c.max_batch = B; c.max_seq = L;
    mb_bert_engine* e = nullptr;
    mb_xlnet_engine* ex = nullptr;
    size_t n, nd, wsb, shb, she;
    if (xl) {
        if (!graph || dp) { fprintf(stderr, "--model xlnet: the single-call step only (--graph 1|2, no --dp)\n"); return 1; }
        mb_xlnet_config xc = {};
        xc.vocab_size = 32000; xc.d_model = 768; xc.n_layer = layers; xc.n_head = 12; xc.d_inner = 3072; xc.num_labels = 1;
        xc.visual_dim = V; xc.acoustic_dim = A; xc.injection_index = 1;
        xc.layer_norm_eps = 1e-12f; xc.mag_layer_norm_eps = 1e-5f; xc.beta_shift = 1.0f;
        xc.dropout = 0.1f; xc.summary_last_dropout = 0.1f; xc.mag_dropout = 0.5f; xc.dtype = dtype; xc.max_batch = B; xc.max_seq = L;
        MCK(mb_xlnet_create(&xc, &ex));
        n = mb_xlnet_param_count(ex); nd = mb_xlnet_decay_count(ex); wsb = mb_xlnet_workspace_bytes(ex);
        mb_xlnet_shadow_range(ex, &shb, &she);
    } else {
        MCK(mb_bert_create(&c, &e));
        n = mb_bert_param_count(e); nd = mb_bert_decay_count(e); wsb = mb_bert_workspace_bytes(e);
        mb_bert_shadow_range(e, &shb, &she);
    }
    float *P, *G, *M, *Vv; void *SH, *WS;
    HCK(hipMalloc(&P, n * 4)); HCK(hipMalloc(&G, n * 4)); HCK(hipMalloc(&M, n * 4)); HCK(hipMalloc(&Vv, n * 4));
    HCK(hipMalloc(&SH, n * 2)); HCK(hipMalloc(&WS, wsb));
    HCK(hipMemset(G, 0, n * 4)); HCK(hipMemset(M, 0, n * 4)); HCK(hipMemset(Vv, 0, n * 4));
    {   // init law of the reference (N(0, 0.02), biases 0, LayerNorm 1/0)
        std::vector<float> h(n, 0.f);
        char name[160]; size_t off, numel; int nd_, dec; int64_t shp[4];
        const int ntens = xl ? mb_xlnet_num_tensors(ex) : mb_bert_num_tensors(e);
        for (int i = 0; i < ntens; ++i) {
            if (xl) MCK(mb_xlnet_tensor_info(ex, i, name, 160, &off, &numel, &nd_, shp, &dec));
            else MCK(mb_bert_tensor_info(e, i, name, 160, &off, &numel, &nd_, shp, &dec));
            const std::string s = name;
            const bool ln = s.find("LayerNorm") != std::string::npos || s.find("layer_norm") != std::string::npos;
            const bool bias = s.size() > 5 && s.compare(s.size() - 5, 5, ".bias") == 0;
            for (size_t j = 0; j < numel; ++j) h[off + j] = ln ? (bias ? 0.f : 1.f) : (bias ? 0.f : 0.02f * nrand());
        }
        HCK(hipMemcpy(P, h.data(), n * 4, hipMemcpyHostToDevice));
    }
    hipStream_t st; HCK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    if (xl) { MCK(mb_xlnet_bind(ex, P, G, dtype == MB_DT_BF16 ? SH : nullptr, WS, wsb)); MCK(mb_xlnet_sync_weights(ex, st)); }
    else { MCK(mb_bert_bind(e, P, G, dtype == MB_DT_BF16 ? SH : nullptr, WS, wsb)); MCK(mb_bert_sync_weights(e, st)); }

    // synthetic batches (host pinned + device resident)
    const size_t T = (size_t)B * L;
    const size_t bytes = T * 8 * 3 + T * V * 4 + T * A * 4 + (size_t)B * 4;
    std::vector<char*> hb(nbatch), db(nbatch);
    auto view = [&](char* p) { Batch b; b.ids = (int64_t*)p; b.seg = b.ids + T; b.mask = b.seg + T; b.vis = (float*)(b.mask + T);
                               b.aco = b.vis + T * V; b.lab = b.aco + T * A; return b; };
    for (int k = 0; k < nbatch; ++k) {
        HCK(hipHostMalloc((void**)&hb[k], bytes)); HCK(hipMalloc((void**)&db[k], bytes));
        Batch b = view(hb[k]);
        memset(hb[k], 0, bytes);
        for (int s = 0; s < B; ++s) {
            const int nw = 5 + (int)(rnd() % (uint64_t)(L - 2 - 5 + 1));
            b.ids[s * L] = 101; b.mask[s * L] = 1;
            for (int t = 1; t <= nw; ++t) {
                b.ids[s * L + t] = 1000 + (int64_t)(rnd() % 29522); b.mask[s * L + t] = 1;
                for (int j = 0; j < V; ++j) b.vis[((size_t)s * L + t) * V + j] = nrand();
                for (int j = 0; j < A; ++j) b.aco[((size_t)s * L + t) * A + j] = nrand();
            }
            b.ids[s * L + nw + 1] = 102; b.mask[s * L + nw + 1] = 1;
            b.lab[s] = 6.f * urand() - 3.f;
        }
        HCK(hipMemcpy(db[k], hb[k], bytes, hipMemcpyHostToDevice));
    }
    float *logits, *loss;
    HCK(hipMalloc(&logits, (size_t)B * 4)); HCK(hipMalloc(&loss, 8)); HCK(hipMemset(loss, 0, 8));

    mb_comm* comm = nullptr;
    if (dp) {
        if (!graph) { fprintf(stderr, "--dp 1 needs --graph 1|2\n"); return 1; }
        char id[128];
        MCK(mb_comm_unique_id(id));
        int rc = mb_comm_create_rccl(id, 0, 1, &comm);
        if (rc) { fprintf(stderr, "mb_comm_create_rccl: %d %s\n", rc, mb_comm_last_error()); return 3; }
        const int vocab = sparse ? c.vocab_size : 0;
        const size_t sb = mb_comm_scratch_bytes(1, wire, n, vocab, c.hidden_size, B * L);
        void* scratch; HCK(hipMalloc(&scratch, sb));
        MCK(mb_comm_bind_scratch(comm, scratch, sb, wire, n, vocab, c.hidden_size, B * L));
        MCK(mb_comm_set_timing(comm, timing));
        if (shard) { MCK(mb_comm_set_sharding(comm, 1)); if (!mb_comm_sharding(comm)) fprintf(stderr, "--shard 1: a one-rank group shards nothing without MB_DP_SHARD_FORCE=1\n"); }
        HCK(hipDeviceSynchronize());
    }
    const float lr = 1e-5f, b1 = 0.9f, b2 = 0.999f, eps = 1e-6f, wd = 0.01f;
    int t_opt = 0;
    auto step = [&](int i) {
        char* src = db[i % nbatch];
        if (h2d == 1) HCK(hipMemcpyAsync(src, hb[i % nbatch], bytes, hipMemcpyHostToDevice, st));     // copy engine, same stream
        if (h2d == 2) src = hb[i % nbatch];       // zero-copy: the step prologue gathers the batch out of pinned host memory (--graph 1|2)
        Batch b = view(src);
        ++t_opt;
        if (graph && dp) {
            MCK(mb_bert_train_step_dp(e, b.ids, b.vis, b.aco, b.mask, b.seg, b.lab, B, L, 1234, (uint64_t)t_opt, logits, loss, loss + 1,
                                      M, Vv, lr, b1, b2, eps, wd, t_opt, 1, 1.0f, 1.0f, graph, st, comm));
            return;
        }
        if (xl) {
            MCK(mb_xlnet_train_step(ex, b.ids, b.vis, b.aco, b.mask, b.seg, b.lab, B, L, 1234, (uint64_t)t_opt, logits, loss, loss + 1,
                                    M, Vv, lr, b1, b2, eps, wd, t_opt, 1, 1.0f, 1.0f, graph, st));
            return;
        }
        if (graph) {
            MCK(mb_bert_train_step(e, b.ids, b.vis, b.aco, b.mask, b.seg, b.lab, B, L, 1234, (uint64_t)t_opt, logits, loss, loss + 1,
                                   M, Vv, lr, b1, b2, eps, wd, t_opt, 1, 1.0f, 1.0f, graph, st));
            return;
        }
        MCK(mb_bert_forward(e, b.ids, b.vis, b.aco, b.mask, b.seg, b.lab, B, L, 1, 1234, (uint64_t)t_opt, logits, loss, loss + 1, st));
        MCK(mb_bert_backward(e, nullptr, b.lab, 1.0f, 0, layers + 2, st));
        MCK(mb_adamw_step(P, G, M, Vv, dtype == MB_DT_BF16 ? SH : nullptr, nd, nd, shb, she, lr, b1, b2, eps, wd, t_opt, 1, 1.0f, 1, st));
        MCK(mb_adamw_step(P + nd, G + nd, M + nd, Vv + nd, nullptr, n - nd, 0, 0, 0, lr, b1, b2, eps, 0.f, t_opt, 1, 1.0f, 1, st));
    };
    for (int i = 0; i < warmup; ++i) step(i);
    HCK(hipStreamSynchronize(st));
    hipEvent_t e0, e1; HCK(hipEventCreate(&e0)); HCK(hipEventCreate(&e1));
    HCK(hipEventRecord(e0, st));
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < steps; ++i) step(warmup + i);
    auto t1 = std::chrono::steady_clock::now();
    HCK(hipEventRecord(e1, st));
    HCK(hipEventSynchronize(e1));
    auto t2 = std::chrono::steady_clock::now();
    float ms = 0.f; HCK(hipEventElapsedTime(&ms, e0, e1));
    float hl[2]; HCK(hipMemcpy(hl, loss, 8, hipMemcpyDeviceToHost));
    const double host_ms = std::chrono::duration<double, std::milli>(t1 - t0).count() / steps;
    const double wall_ms = std::chrono::duration<double, std::milli>(t2 - t0).count() / steps;
    printf("step_bench %sdtype=%s B=%d L=%d V=%d layers=%d graph=%d h2d=%d : %.3f ms/step (events) %.3f ms/step (wall) host-enqueue %.3f ms/step "
           "%.1f samples/s last-loss %.4f mean-loss %.4f\n",
           xl ? "model=xlnet " : "", dtype == MB_DT_BF16 ? "bf16" : "fp32", B, L, V, layers, graph, h2d, ms / steps, wall_ms, host_ms, B * 1e3 / (ms / steps), hl[0],
           hl[1] / (steps + warmup));
    if (comm) {
        float ex = 0.f; size_t pieces = 0, cbytes = 0;
        MCK(mb_comm_exposed_ms(comm, &ex)); MCK(mb_comm_stats(comm, &pieces, &cbytes));
        printf("step_bench dp: 1-rank RCCL, wire=%s sparse=%d shard=%d : %zu collectives / %.1f MB per step, comm_exposed %.4f ms (last step)\n",
               wire == MB_DT_BF16 ? "bf16" : "fp32", sparse, mb_comm_sharding(comm), pieces, cbytes * 1e-6, ex);
        mb_comm_destroy(comm);
    }
    if (xl) mb_xlnet_destroy(ex); else mb_bert_destroy(e);
    return 0;
}
