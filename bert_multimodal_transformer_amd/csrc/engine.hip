// Native step executor for MAG_BertForSequenceClassification (/root/reference/bert.py:240-324 -> :76-237 ->
// /root/reference/modeling.py:25-51) and the C ABI declared in include/magbert_hip.h.
//
// One call = one pass: every kernel of the forward (or of a backward stage range) is enqueued on the caller's
// stream from C++, so the per-step host cost is ~100-300 plain launches and zero Python.  The engine owns no
// device memory: parameters / gradients / bf16 operand shadow / workspace are caller buffers (torch tensors).
//
// Flat parameter layout (floats; every tensor 256-byte aligned; names = reference state-dict keys):
//   [ weight-decay group | no-decay group ]   (multimodal_driver.py:329-343: "bias", "LayerNorm.*" do not decay)
//   decay group   : per layer {query,key,value (contiguous = fused [3H][H] QKV), attention.output.dense,
//                   intermediate.dense, output.dense}.weight ; pooler.dense.weight   <- bf16 shadow range
//                   embeddings.{word,position,token_type}_embeddings.weight ; MAG.{W_hv,W_ha,W_v,W_a}.weight ;
//                   classifier.weight
//   no-decay group: per layer {q,k,v bias (contiguous [3H]), attn.out bias, LN1 w/b, inter bias, out bias, LN2 w/b};
//                   embeddings.LayerNorm ; pooler bias ; MAG biases + LayerNorm ; classifier.bias
#include "engine_common.h"
#include "comm.h"

// ================================================================================================ engine
struct LayerOff { size_t wqkv, wo, w1, w2, bqkv, bo, ln1w, ln1b, b1, b2, ln2w, ln2b; };
struct LayerWs { size_t qkv, ctx, s1, st1, y1, u, g, s2, st2; };

struct mb_bert_engine : StepMixin {
    mb_bert_config c;
    std::vector<TensorInfo> tensors;
    std::vector<LayerOff> lo;
    size_t word, pos, type, emb_lnw, emb_lnb, wp, bp, wc, bc;
    size_t mag_whv, mag_wha, mag_wv, mag_wa, mag_bhv, mag_bha, mag_bv, mag_ba, mag_lnw, mag_lnb;
    size_t n_params, n_decay, sh_begin, sh_end;
    // workspace
    MagWs mw;
    size_t ws_mag, ws_emb, ws_emb_st, ws_head_z, ws_head_pooled, ws_logits;
    std::vector<size_t> ws_x;
    std::vector<LayerWs> lw;
    size_t ws_ds[2], ws_dzd[2], ws_ds2[2], ws_dzd2[2], ws_du[2], ws_dqkv[2];   // dY operands of the wgrads: ping-pong by layer parity
    size_t ws_dxa, ws_dxb, ws_dctx, ws_dsum, ws_dz, ws_lnp_a, ws_lnp_b;
    size_t lnp_stride = 0;         // floats per layer in each of the two LayerNorm partial buffers
    int lnp_nblk = 0;              // slabs per layer written by the current backward
    size_t ws_ids, ws_seg, ws_mask, ws_labels;    // (inputs are caller pointers; kept for the backward)
    size_t ws_bytes;
    // bound buffers
    float* P = nullptr; float* G = nullptr; char* SH = nullptr; char* ws = nullptr;
    // state of the last forward
    const int64_t* ids = nullptr; const int64_t* seg = nullptr; const int64_t* mask = nullptr;
    int B = 0, L = 0, training = 0;
    int padT = -1;                 // token count whose pad rows [T, Tp) are currently known to be zero
    // weight-gradient GEMMs run on an internal side stream, concurrently with the dgrad chain of the same layer
    hipStream_t side = nullptr;
    std::vector<hipEvent_t> evs;      // 5 events per encoder stage (4 forks + 1 join), never reused within a backward
    int overlap_wgrad = 0;         // MB_OVERLAP_WGRAD=1: weight-gradient launches on the internal side stream (round-1 default; measured equal
                                   // to the in-line grouped launch, which keeps the step a single-stream sequence -- and its hipGraph a fast one)
    // MB_ADAMW_OVERLAP=C: the single-call step forks the optimizer of every finished chunk of C layers onto this stream (enqueue_step)
    int opt_chunk = 0;
    int prefetch = 1;              // MB_PREFETCH=0: the LayerNorm kernels do not touch the next GEMMs' weights (common.h Prefetch).  Touching
                                   // ACTIVATIONS the same way (GELU output for the weight gradient, saved q | k | v for the attention backward)
                                   // was measured +9 / +15 us per step and is not in the code (profiles/r03_prefetch_ab2.txt)
    int pf_qkv = 0;                // MB_PF_QKV=64|128: ln_bwd(LN1)'s left-over prefetch loads touch the saved q | k | v | context rows, one per 64 / 128 bytes
    hipStream_t opt_side = nullptr;
    std::vector<hipEvent_t> opt_ev;
    // EXPERIMENT MB_ADAMW_IN_WGRAD=1 (kernels.h: EPI_WGRAD_ADAM): in a single-process single-call step whose gradient buffer is known-zero,
    // the layers' grouped weight-gradient launches update the parameters themselves and the optimizer sweep skips that range.
    int adam_in_wgrad = 0;
    float* fuse_m = nullptr; float* fuse_v = nullptr;      // Adam moments of the step being enqueued, when the fusion applies to it
    // MB_ADAMW_RIDE=1 (kernels.h AdamRide): in a single-process single-call step the grouped weight-gradient launch of layer l carries
    // the optimizer update of layer l+1's GEMM weights (whose gradients the launch before completed) as extra workgroups in the slots
    // its tiles leave empty (256 x 128 tiles: 40 of 256 CUs; 128 x 128: 80 of 512 slots); the sweep at the end skips those layers.
    int adam_ride = 1, ride_blocks = 0;
    float* ride_m = nullptr; float* ride_v = nullptr;       // Adam moments of the step being enqueued, when riders apply to it
    size_t ride_cursor = 0;                                 // the riders of the step being enqueued have taken [ride_cursor, wp) of the decay slab
    long ride_params = 0;                                   // MB_ADAMW_RIDE_PARAMS: parameters per launch (0 = by the token count)
    // MB_ADAMW_RIDE_DGRAD >= 1: riders also in the two 64 x 64 dgrad launches of a layer (ffn1, qkv: 456 tiles in 768 block slots at T = 2400;
    // kernels.h gemm_nn_ride_launch), 2 (default): and in the 128 x 128 ffn2 dgrad (456 tiles in 512 slots: 56 CUs hold one tile).
    // _PARAMS: parameters per 64 x 64 launch (0 = by the launch's FLOPs), _DGELU_PARAMS: per ffn2 launch (0 = 14,336 per free slot),
    // _BLOCKS: rider workgroups (0 = every free slot).  Same box: 3.422 ms off | 3.411 (1) | 3.401 (2)  (profiles/r06_adamw_ride_dgrad.txt)
    int ride_dgrad = 2, ride_dgrad_blocks = 0;
    int ride_attn = 1, ride_attn_blocks = 0;                // MB_ADAMW_RIDE_ATTN: riders in the attention backward launch (_BLOCKS, _PARAMS: as above)
    long ride_attn_params = 0;
    long ride_dgrad_params = 0, ride_dgelu_params = 0;      // (MB_ADAMW_RIDE_DGRAD=2: also the ffn2 dgrad; _DGELU_PARAMS: parameters per such launch)
    int group_wgrad = 256;         // MB_GROUP_WGRAD: tile of the per-layer grouped wgrad launch (64 | 128 | 256 = 256 x 128 ping-pong), 0 = four launches
    bool grouped = false;          // the layer's four weight gradients are ONE launch
    bool deferred = false;         // ... on the side stream, joined one stage later (MB_OVERLAP_WGRAD=0: on the caller's stream, in line)
    float* attn_out = nullptr;     // mb_bert_set_attention_output: [num_layers][B][nh][L][L] fp32, filled by the next forwards
    const float* head_mask = nullptr;   // mb_bert_set_head_mask: [num_layers][num_heads] fp32 (caller-owned device memory)
    const float* emb_in = nullptr;      // mb_bert_set_inputs_embeds: [B*L][H] fp32 word embeddings given instead of input_ids
    const int64_t* pos_ids = nullptr;   // mb_bert_set_position_ids: [B*L] rows of the position table (null: arange(L), the default)
    bool ran_forward = false;

    bool ws_zeroed = false;
    uint64_t seed = 0, step = 0;
    float* logits = nullptr;
    // (whole-step machinery: StepMixin -- staging buffers, device-resident step state, graph cache)

    size_t add(const std::string& name, std::vector<int64_t> shape, int decay, size_t& cursor) {
        TensorInfo t;
        t.name = name; t.ndim = (int)shape.size(); t.decay = decay; t.numel = 1;
        for (int i = 0; i < 4; ++i) t.shape[i] = i < t.ndim ? shape[i] : 1;
        for (auto s : shape) t.numel *= (size_t)s;
        t.off = cursor;
        cursor = align_up(cursor + t.numel, 64);
        tensors.push_back(t);
        return t.off;
    }
    const void* W(size_t off) const {   // GEMM operand view of a weight (bf16 shadow in perf mode, master in fp32 mode)
        return c.dtype == DT_BF16 ? (const void*)(SH + off * 2) : (const void*)(P + off);
    }
    DropKey key(uint32_t site, float p) const { return step_key(ws, training != 0, seed, step, site, p); }
};

static void build_layout(mb_bert_engine* e) {
    const mb_bert_config& c = e->c;
    const int64_t H = c.hidden_size, I = c.intermediate_size, V = c.visual_dim, A = c.acoustic_dim;
    size_t cur = 0;
    e->lo.resize(c.num_layers);
    char buf[128];
    auto nm = [&](int l, const char* s) { snprintf(buf, sizeof buf, "bert.encoder.layer.%d.%s", l, s); return std::string(buf); };
    e->sh_begin = 0;
    for (int l = 0; l < c.num_layers; ++l) {
        LayerOff& o = e->lo[l];
        o.wqkv = e->add(nm(l, "attention.self.query.weight"), {H, H}, 1, cur);
        e->add(nm(l, "attention.self.key.weight"), {H, H}, 1, cur);
        e->add(nm(l, "attention.self.value.weight"), {H, H}, 1, cur);
        o.wo = e->add(nm(l, "attention.output.dense.weight"), {H, H}, 1, cur);
        o.w1 = e->add(nm(l, "intermediate.dense.weight"), {I, H}, 1, cur);
        o.w2 = e->add(nm(l, "output.dense.weight"), {H, I}, 1, cur);
    }
    e->wp = e->add("bert.pooler.dense.weight", {H, H}, 1, cur);
    e->sh_end = cur;
    e->word = e->add("bert.embeddings.word_embeddings.weight", {c.vocab_size, H}, 1, cur);
    e->pos = e->add("bert.embeddings.position_embeddings.weight", {c.max_position, H}, 1, cur);
    e->type = e->add("bert.embeddings.token_type_embeddings.weight", {c.type_vocab, H}, 1, cur);
    e->mag_whv = e->add("bert.MAG.W_hv.weight", {H, V + H}, 1, cur);
    e->mag_wha = e->add("bert.MAG.W_ha.weight", {H, A + H}, 1, cur);
    e->mag_wv = e->add("bert.MAG.W_v.weight", {H, V}, 1, cur);
    e->mag_wa = e->add("bert.MAG.W_a.weight", {H, A}, 1, cur);
    e->wc = e->add("classifier.weight", {c.num_labels, H}, 1, cur);
    e->n_decay = cur;
    for (int l = 0; l < c.num_layers; ++l) {
        LayerOff& o = e->lo[l];
        o.bqkv = e->add(nm(l, "attention.self.query.bias"), {H}, 0, cur);
        e->add(nm(l, "attention.self.key.bias"), {H}, 0, cur);
        e->add(nm(l, "attention.self.value.bias"), {H}, 0, cur);
        o.bo = e->add(nm(l, "attention.output.dense.bias"), {H}, 0, cur);
        o.ln1w = e->add(nm(l, "attention.output.LayerNorm.weight"), {H}, 0, cur);
        o.ln1b = e->add(nm(l, "attention.output.LayerNorm.bias"), {H}, 0, cur);
        o.b1 = e->add(nm(l, "intermediate.dense.bias"), {I}, 0, cur);
        o.b2 = e->add(nm(l, "output.dense.bias"), {H}, 0, cur);
        o.ln2w = e->add(nm(l, "output.LayerNorm.weight"), {H}, 0, cur);
        o.ln2b = e->add(nm(l, "output.LayerNorm.bias"), {H}, 0, cur);
    }
    e->emb_lnw = e->add("bert.embeddings.LayerNorm.weight", {H}, 0, cur);
    e->emb_lnb = e->add("bert.embeddings.LayerNorm.bias", {H}, 0, cur);
    e->bp = e->add("bert.pooler.dense.bias", {H}, 0, cur);
    e->mag_bhv = e->add("bert.MAG.W_hv.bias", {H}, 0, cur);
    e->mag_bha = e->add("bert.MAG.W_ha.bias", {H}, 0, cur);
    e->mag_bv = e->add("bert.MAG.W_v.bias", {H}, 0, cur);
    e->mag_ba = e->add("bert.MAG.W_a.bias", {H}, 0, cur);
    e->mag_lnw = e->add("bert.MAG.LayerNorm.weight", {H}, 0, cur);
    e->mag_lnb = e->add("bert.MAG.LayerNorm.bias", {H}, 0, cur);
    e->bc = e->add("classifier.bias", {c.num_labels}, 0, cur);
    e->n_params = cur;

    // ---- workspace
    const size_t es = esize(c.dtype);
    const size_t T = align_up((size_t)c.max_batch * c.max_seq, 64);   // token rows padded to the GEMM k-tile
    Carver w;
    e->mw.init(c.dtype, (int)T, (int)H, (int)V, (int)A);
    e->ws_mag = w.take(e->mw.bytes);
    {   // MB_PROLOGUE_PACK=0: the step prologue stages the fp32 modality tensors and the forward packs them (two more launches)
        const char* pv = getenv("MB_PROLOGUE_PACK");
        e->pk_enable = !(pv && atoi(pv) == 0);
        e->pk_vis = e->ws_mag + e->mw.vp; e->pk_aco = e->ws_mag + e->mw.ap; e->pk_Vp = e->mw.Vp; e->pk_Ap = e->mw.Ap; e->pk_dtype = c.dtype;
        const char* pw = getenv("MB_PROLOGUE_PACKW");
        e->pkw_enable = !(pw && atoi(pw) == 0);
    }
    e->ws_emb = w.take(T * H * es);
    e->ws_emb_st = w.take(2 * T * 4);
    e->ws_x.resize(c.num_layers + 1);
    for (int l = 0; l <= c.num_layers; ++l) e->ws_x[l] = w.take(T * H * es);
    e->lw.resize(c.num_layers);
    for (int l = 0; l < c.num_layers; ++l) {
        LayerWs& x = e->lw[l];
        x.qkv = w.take(T * 3 * H * es); x.ctx = w.take(T * H * es); x.s1 = w.take(T * H * es); x.st1 = w.take(2 * T * 4);
        x.y1 = w.take(T * H * es); x.u = w.take(T * I * es); x.g = w.take(T * I * es); x.s2 = w.take(T * H * es);
        x.st2 = w.take(2 * T * 4);
    }
    e->ws_head_z = w.take((size_t)c.max_batch * H * 4);
    e->ws_head_pooled = w.take((size_t)c.max_batch * H * 4);
    e->ws_logits = w.take((size_t)c.max_batch * c.num_labels * 4);
    e->ws_dxa = w.take(T * H * es); e->ws_dxb = w.take(T * H * es);
    for (int k = 0; k < 2; ++k) {
        e->ws_ds[k] = w.take(T * H * es); e->ws_dzd[k] = w.take(T * H * es); e->ws_ds2[k] = w.take(T * H * es);
        e->ws_dzd2[k] = w.take(T * H * es); e->ws_du[k] = w.take(T * I * es); e->ws_dqkv[k] = w.take(T * 3 * H * es);
    }
    e->ws_dctx = w.take(T * H * es); e->ws_dsum = w.take(T * H * 4); e->ws_dz = w.take((size_t)c.max_batch * H * es);
    // LayerNorm partial slabs of EVERY layer (2 x 2.8 MB per layer at T = 2400): the single-call step reduces them in one launch
    e->lnp_stride = ln_partials_floats((int)T, (int)H);
    {   // the one-launch embedding backward writes max_seq * ceil(max_batch / 8) slabs (more than ceil(T / 8) for batches not a multiple of 8)
        const size_t emb = (size_t)c.max_seq * ((c.max_batch + 7) / 8) * 3 * H;
        if (emb > e->lnp_stride) e->lnp_stride = emb;
    }
    e->ws_lnp_a = w.take(e->lnp_stride * 4 * (c.num_layers + 2)); e->ws_lnp_b = w.take(e->lnp_stride * 4 * (c.num_layers + 2));     // (+1: MAG's gate, +1: the embeddings)
    e->carve_step(w, T, (int)V, (int)A, c.max_batch, c.num_labels, SITE_LAYER0 + 4 * c.num_layers);
    e->idcnt_off = w.take((size_t)c.vocab_size * 4);          // token-id occurrence table of the single-call step (MB_EMBED_UNIQUE=0: off)
    { const char* uv = getenv("MB_EMBED_UNIQUE"); e->idcnt_enable = !(uv && atoi(uv) == 0); }
    if (e->deterministic) {          // shadow accumulator of everything behind the layers' GEMM weights (those have ONE writer per element)
        e->det_begin = e->wp; e->det_end = e->n_params;
        e->ws_det = w.take((e->det_end - e->det_begin) * sizeof(long long));
    }
    e->ws_bytes = w.off;
}

// the internal side stream + its fork / join events (created once, outside any stream capture)
static int ensure_side(mb_bert_engine* e) {
    if (e->side || !e->overlap_wgrad) return MB_OK;
    // MB_SIDE_PRIORITY=1: lowest dispatch priority for the weight-gradient stream (the dgrad chain is the critical
    // path).  Measured: no effect -- resident wgrad blocks keep their LDS slots for a whole K = T loop, priority
    // only orders NEW workgroups -- so the default stays the normal priority.
    int least = 0, greatest = 0;
    CK((int)hipDeviceGetStreamPriorityRange(&least, &greatest));
    const char* pv = getenv("MB_SIDE_PRIORITY");
    const int prio = (pv && atoi(pv) != 0) ? least : 0;
    CK((int)hipStreamCreateWithPriority(&e->side, hipStreamNonBlocking, prio));
    e->evs.assign((size_t)e->c.num_layers * 5, nullptr);
    for (auto& ev : e->evs) CK((int)hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    return MB_OK;
}

// State the next pass relies on but that is not part of the pass itself (kept out of captured step graphs): outstanding
// side-stream work of an unfinished backward, the one-time clearing of the workspace, zero pad rows for this token count.
static int prepare_pass(mb_bert_engine* e, int T, hipStream_t st) {
    const mb_bert_config& c = e->c;
    const int H = c.hidden_size, I = c.intermediate_size, dt = c.dtype;
    char* ws = e->ws;
    const int Tp = (int)align_up((size_t)T, 64);
    if (e->deferred && e->side)       // a backward that was not run to its last stage may still have weight-gradient GEMMs reading activations
        for (size_t l = 0; l < 2 && l * 5 + 4 < e->evs.size(); ++l) CK((int)hipStreamWaitEvent(st, e->evs[l * 5 + 4], 0));
    if (!e->ws_zeroed) { CK((int)hipMemsetAsync(ws, 0, e->ws_bytes, st)); e->ws_zeroed = true; e->padT = T; }
    if (e->padT != T && Tp > T) {
        // a different batch shape ran before: rows [T, Tp) of every buffer that feeds a wgrad as the k-major operand
        // may hold stale tokens -> clear them (kernels never write rows >= T)
        const size_t es = esize(dt);
        auto zp = [&](size_t off, size_t cols) {
            return (int)hipMemsetAsync(ws + off + (size_t)T * cols * es, 0, (size_t)(Tp - T) * cols * es, st);
        };
        CK(zp(e->ws_emb, H));
        CK(mag_clear_pad_rows(dt, ws + e->ws_mag, e->mw, T, H, st));
        for (int k = 0; k < 2; ++k) {
            CK(zp(e->ws_ds[k], H)); CK(zp(e->ws_dzd[k], H)); CK(zp(e->ws_ds2[k], H)); CK(zp(e->ws_dzd2[k], H));
            CK(zp(e->ws_du[k], I)); CK(zp(e->ws_dqkv[k], 3 * H));
        }
        for (int l = 0; l <= c.num_layers; ++l) CK(zp(e->ws_x[l], H));
        for (int l = 0; l < c.num_layers; ++l) { CK(zp(e->lw[l].ctx, H)); CK(zp(e->lw[l].y1, H)); CK(zp(e->lw[l].g, I)); }
    }
    e->padT = T;
    return MB_OK;
}

extern "C" {

const char* mb_error_string(int code) {
    switch (code) {
        case MB_OK: return "ok";
        case MB_ERR_SHAPE: return "magbert: unsupported shape or alignment";
        case MB_ERR_MODE: return "magbert: unsupported layout/epilogue combination";
        case MB_ERR_DTYPE: return "magbert: unsupported dtype";
        case MB_ERR_ARG: return "magbert: invalid argument";
        case MB_ERR_COMM: return "magbert: gradient exchange unavailable (mb_comm_last_error())";
        default:
            if (code >= 2000 && code < 2100) return "magbert: RCCL call failed (mb_comm_last_error())"; return hipGetErrorString((hipError_t)code);
    }
}
int mb_version(void) { return 100; }

void mb_make_dropkey(uint64_t seed, uint64_t step, uint32_t site, float p, mb_dropkey* out) {
    DropKey k = make_key(seed, step, site, p);
    out->k0 = k.k0; out->k1 = k.k1; out->thresh = k.thresh; out->scale = k.scale;
}

int mb_gemm(int dtype, int layout, int epilogue, int M, int N, int K, const void* A, int lda, const void* B, int ldb,
            void* C, int ldc, void* C2, float* Cf, const float* bias, const void* R, int ldr, float alpha,
            const mb_dropkey* drop, int splits, int tile, void* stream) {
    GemmArgs a = {};
    a.A = A; a.B = B; a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.C = C; a.ldc = ldc; a.C2 = C2; a.Cf = Cf;
    a.bias = bias; a.R = R; a.ldr = ldr; a.alpha = alpha; a.drop = dk(drop); a.kchunk = K; a.colsum = nullptr;
    if (epilogue == EPI_DGELU) { a.colsum = Cf; a.Cf = nullptr; }       // epilogue 4: the fp32 pointer is the fused bias gradient
    return gemm_launch(dtype, layout, epilogue, a, splits, tile, (hipStream_t)stream);
}

int mb_debug_gemm_trace(unsigned long long* host_out, int max_blocks) { return gemm_trace_fetch(host_out, max_blocks); }
int mb_debug_attention_trace(unsigned long long* host_out, int max_blocks) { return attention_trace_fetch(host_out, max_blocks); }
int mb_gemm_grouped_wgrad(int dtype, int count, const int* M, const int* N, int K, const void* const* dY, const int* ldy,
                          const void* const* X, const int* ldx, float* const* dW, const int* ldw, int tile, void* stream) {
    if (count < 1 || count > MB_MAX_GROUP || !M || !N || !dY || !X || !dW || !ldy || !ldx || !ldw) return MB_ERR_ARG;
    GemmArgs g[MB_MAX_GROUP];
    for (int i = 0; i < count; ++i) g[i] = wgrad_args(M[i], N[i], K, dY[i], ldy[i], X[i], ldx[i], dW[i], ldw[i]);
    if (!gemm_grouped_tn_ok(dtype, g, count, tile)) return MB_ERR_SHAPE;
    return gemm_grouped_tn_launch(dtype, g, count, tile, (hipStream_t)stream);
}

int mb_narrow(int dtype, const float* src, void* dst, size_t n, void* stream) {
    if (!src || !dst) return MB_ERR_ARG;
    if (n % 4) return MB_ERR_SHAPE;
    return convert(dtype, src, dst, n, (hipStream_t)stream);
}
int mb_widen(int dtype, const void* src, float* dst, size_t n, void* stream) {
    if (!src || !dst) return MB_ERR_ARG;
    return widen(dtype, src, dst, n, (hipStream_t)stream);
}

int mb_layernorm_forward(int dtype, const void* x, const float* gamma, const float* beta, float eps, void* y, float* mean,
                         float* rstd, int rows, int H, const mb_dropkey* drop, void* stream) {
    return ln_forward(dtype, x, gamma, beta, eps, y, mean, rstd, rows, H, dk(drop), (hipStream_t)stream);
}
int mb_layernorm_backward(int dtype, const void* dy, const void* x, const float* gamma, const float* mean,
                          const float* rstd, void* dx, void* dx_drop, float* dgamma, float* dbeta, float* dbias, int rows,
                          int H, const mb_dropkey* drop_out, const mb_dropkey* drop_in, void* stream) {
    return ln_backward(dtype, dy, x, gamma, mean, rstd, dx, dx_drop, dgamma, dbeta, dbias, rows, H, dk(drop_out),
                       dk(drop_in), (hipStream_t)stream);
}
int mb_embed_forward(int dtype, const int64_t* ids, const int64_t* seg, const float* word, const float* pos,
                     const float* type, const float* gamma, const float* beta, float eps, void* out, float* mean,
                     float* rstd, int B, int L, int H, const mb_dropkey* drop, void* stream) {
    return embed_ln_forward(dtype, ids, seg, word, pos, type, gamma, beta, eps, out, mean, rstd, B, L, H, dk(drop),
                            (hipStream_t)stream);
}
int mb_embed_backward(int dtype, const void* dout, const int64_t* ids, const int64_t* seg, const float* word,
                      const float* pos, const float* type, const float* gamma, const float* mean, const float* rstd,
                      float* dsum_ws, float* dword, float* dpos, float* dtype_, float* dgamma, float* dbeta, int B, int L,
                      int H, int pad_id, const mb_dropkey* drop, void* stream) {
    return embed_ln_backward(dtype, dout, ids, seg, word, pos, type, gamma, mean, rstd, dsum_ws, dword, dpos, dtype_,
                             dgamma, dbeta, B, L, H, pad_id, dk(drop), (hipStream_t)stream);
}
int mb_attention_forward(int dtype, const void* qkv, const int64_t* mask, void* ctx, int B, int L, int nh,
                         const mb_dropkey* drop, void* stream) {
    return attention_forward(dtype, qkv, mask, ctx, B, L, nh, dk(drop), (hipStream_t)stream);
}
int mb_attention_backward(int dtype, const void* qkv, const int64_t* mask, const void* dctx, void* dqkv, int B, int L,
                          int nh, const mb_dropkey* drop, void* stream) {
    // MB_ATTN_TRACE=1 (tools/attn_bench): the traced launch also produces the fused QKV bias gradient, as inside the engine
    static float* trace_dbias = nullptr;
    static int trace_on = -1;
    if (trace_on < 0) {
        const char* v = getenv("MB_ATTN_TRACE");
        trace_on = v ? atoi(v) : 0;
        if (trace_on && (hipMalloc(&trace_dbias, (size_t)3 * 64 * 64 * sizeof(float)) != hipSuccess ||
                         hipMemset(trace_dbias, 0, (size_t)3 * 64 * 64 * sizeof(float)) != hipSuccess)) trace_dbias = nullptr;
    }
    return attention_backward(dtype, qkv, mask, nullptr, dctx, dqkv, (trace_on && nh <= 64) ? trace_dbias : nullptr, B, L, nh, dk(drop),
                              (hipStream_t)stream);
}

size_t mb_mag_workspace_bytes(int dtype, int T, int H, int V, int A) {
    MagWs w;
    w.init(dtype, T, H, V, A);
    return w.bytes;
}
int mb_mag_forward(int dtype, const void* text, const float* visual, const float* acoustic, const float* W_hv,
                   const float* b_hv, const float* W_ha, const float* b_ha, const float* W_v, const float* b_v,
                   const float* W_a, const float* b_a, const float* ln_w, const float* ln_b, float beta_shift,
                   const mb_dropkey* drop, void* out, void* ws, int T, int H, int V, int A, void* stream) {
    if (H % 256 || H < 256 || H > 1024 || V < 1 || A < 1) return MB_ERR_SHAPE;      // the operator; the engines are built for 768
    MagWs w;
    w.init(dtype, T, H, V, A);
    return mag_fwd_impl(dtype, text, visual, acoustic, W_hv, b_hv, W_ha, b_ha, W_v, b_v, W_a, b_a, ln_w, ln_b, 1e-5f,
                        beta_shift, dk(drop), out, (char*)ws, w, T, H, V, A, true, (hipStream_t)stream);
}
int mb_mag_backward(int dtype, const void* d_out, const void* text, const float* W_hv, const float* b_hv,
                    const float* W_ha, const float* b_ha, const float* W_v, const float* b_v, const float* W_a,
                    const float* b_a, const float* ln_w, float beta_shift, const mb_dropkey* drop, void* ws, void* d_text,
                    float* d_visual, float* d_acoustic, float* dW_hv, float* db_hv, float* dW_ha, float* db_ha,
                    float* dW_v, float* db_v, float* dW_a, float* db_a, float* dln_w, float* dln_b, int T, int H, int V,
                    int A, void* stream) {
    (void)W_hv; (void)W_ha; (void)W_v; (void)W_a;   // the packed copies made by the forward are reused
    if (H % 256 || H < 256 || H > 1024) return MB_ERR_SHAPE;
    MagWs w;
    w.init(dtype, T, H, V, A);
    return mag_bwd_impl(dtype, d_out, text, b_hv, b_ha, b_v, b_a, ln_w, beta_shift, dk(drop), (char*)ws, w, d_text,
                        d_visual, d_acoustic, dW_hv, db_hv, dW_ha, db_ha, dW_v, db_v, dW_a, db_a, dln_w, dln_b, T, H, V, A,
                        false, (hipStream_t)stream);
}

int mb_adamw_step(float* p, float* g, float* m, float* v, void* shadow, size_t n, size_t n_decay, size_t sh_begin,
                  size_t sh_end, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                  int correct_bias, float grad_scale, int zero_grad, void* stream) {
    AdamArgs a;
    a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.weight_decay = weight_decay; a.grad_scale = grad_scale;
    double ss = lr;
    if (correct_bias) ss = (double)lr * sqrt(1.0 - pow((double)beta2, (double)step)) / (1.0 - pow((double)beta1, (double)step));
    a.step_size = (float)ss;
    return adamw_step(p, g, m, v, shadow, n, n_decay, sh_begin, sh_end, a, zero_grad, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------ engine API
int mb_bert_create(const mb_bert_config* cfg, mb_bert_engine** out) {
    if (!cfg || !out) return MB_ERR_ARG;
    if (cfg->hidden_size != 768 || cfg->num_heads * 64 != cfg->hidden_size) return MB_ERR_SHAPE;
    if (cfg->intermediate_size % 128 || cfg->max_seq < 1 || cfg->max_seq > 128 || cfg->max_batch < 1) return MB_ERR_SHAPE;
    if (cfg->max_seq > cfg->max_position || cfg->num_labels < 1) return MB_ERR_SHAPE;
    if (cfg->dtype != DT_F32 && cfg->dtype != DT_BF16) return MB_ERR_DTYPE;
    mb_bert_engine* e = new mb_bert_engine();
    e->c = *cfg;
    if (const char* v = getenv("MB_OVERLAP_WGRAD")) e->overlap_wgrad = atoi(v);
    if (const char* v = getenv("MB_GROUP_WGRAD")) e->group_wgrad = atoi(v);
    if (const char* v = getenv("MB_WGRAD_OVERWRITE")) e->ow_enable = atoi(v);
    if (const char* v = getenv("MB_ADAMW_KEEP")) e->keep_enable = atoi(v);
    if (const char* v = getenv("MB_ADAMW_IN_WGRAD")) e->adam_in_wgrad = atoi(v);
    if (const char* v = getenv("MB_ADAMW_RIDE")) e->adam_ride = atoi(v);
    if (const char* v = getenv("MB_ADAMW_RIDE_BLOCKS")) e->ride_blocks = atoi(v);
    if (const char* v = getenv("MB_ADAMW_RIDE_PARAMS")) e->ride_params = atol(v);
    if (const char* v = getenv("MB_ADAMW_RIDE_DGRAD")) e->ride_dgrad = atoi(v);
    if (const char* v = getenv("MB_ADAMW_RIDE_DGRAD_BLOCKS")) e->ride_dgrad_blocks = atoi(v);
    if (const char* v = getenv("MB_ADAMW_RIDE_DGRAD_PARAMS")) e->ride_dgrad_params = atol(v);
    if (const char* v = getenv("MB_ADAMW_RIDE_DGELU_PARAMS")) e->ride_dgelu_params = atol(v);
    if (const char* v = getenv("MB_ADAMW_RIDE_ATTN")) e->ride_attn = atoi(v);
    if (const char* v = getenv("MB_ADAMW_RIDE_ATTN_BLOCKS")) e->ride_attn_blocks = atoi(v);
    if (const char* v = getenv("MB_ADAMW_RIDE_ATTN_PARAMS")) e->ride_attn_params = atol(v);
    if (const char* v = getenv("MB_DETERMINISTIC")) e->deterministic = atoi(v);
    if (const char* v = getenv("MB_ADAMW_OVERLAP")) e->opt_chunk = atoi(v);
    if (const char* v = getenv("MB_PREFETCH")) e->prefetch = atoi(v);
    if (const char* v = getenv("MB_PF_QKV")) e->pf_qkv = atoi(v);
    if (e->adam_in_wgrad && !getenv("MB_GROUP_WGRAD")) e->group_wgrad = 128;      // (that experiment lives in the 128 x 128 kernel's epilogue)
    // 256 = the 256 x 128 ping-pong tile (gemm_pp.hip; bf16 only): falls back to 128 where it does not divide the layer
    if (e->group_wgrad == 256 && (cfg->dtype != DT_BF16 || cfg->hidden_size % 256 != 0 || cfg->intermediate_size % 256 != 0)) e->group_wgrad = 128;
    e->grouped = (e->group_wgrad == 64 || e->group_wgrad == 128 || e->group_wgrad == 256) && cfg->hidden_size % e->group_wgrad == 0 &&
                 cfg->intermediate_size % e->group_wgrad == 0;
    e->deferred = e->overlap_wgrad && e->grouped;

    build_layout(e);
    // lazy zeroing: the range the grouped launches of all layers store into = every layer's four GEMM weights (contiguous)
    e->ow_covers = e->grouped;
    e->stale_begin = e->lo[0].wqkv; e->stale_end = e->wp;
    *out = e;
    return MB_OK;
}
void mb_bert_destroy(mb_bert_engine* e) {
    if (!e) return;
    if (e->side) hipStreamDestroy(e->side);
    for (auto& ev : e->evs) if (ev) hipEventDestroy(ev);
    if (e->opt_side) hipStreamDestroy(e->opt_side);
    for (auto& ev : e->opt_ev) if (ev) hipEventDestroy(ev);
    e->destroy_prof();
    e->drop_graphs();
    delete e;
}
int mb_bert_num_tensors(const mb_bert_engine* e) { return (int)e->tensors.size(); }
int mb_bert_tensor_info(const mb_bert_engine* e, int i, char* name, int name_cap, size_t* offset, size_t* numel, int* ndim,
                        int64_t* shape4, int* decay) {
    if (i < 0 || i >= (int)e->tensors.size()) return MB_ERR_ARG;
    const TensorInfo& t = e->tensors[i];
    if (name && name_cap > 0) { strncpy(name, t.name.c_str(), name_cap - 1); name[name_cap - 1] = 0; }
    if (offset) *offset = t.off;
    if (numel) *numel = t.numel;
    if (ndim) *ndim = t.ndim;
    if (shape4) for (int k = 0; k < 4; ++k) shape4[k] = t.shape[k];
    if (decay) *decay = t.decay;
    return MB_OK;
}
size_t mb_bert_param_count(const mb_bert_engine* e) { return e->n_params; }
size_t mb_bert_decay_count(const mb_bert_engine* e) { return e->n_decay; }
void mb_bert_shadow_range(const mb_bert_engine* e, size_t* b, size_t* en) { *b = e->sh_begin; *en = e->sh_end; }
size_t mb_bert_workspace_bytes(const mb_bert_engine* e) { return e->ws_bytes; }

int mb_bert_bind(mb_bert_engine* e, float* params, float* grads, void* shadow, void* workspace, size_t ws_bytes) {
    if (!params || !workspace || ws_bytes < e->ws_bytes) return MB_ERR_ARG;
    if (e->c.dtype == DT_BF16 && !shadow) return MB_ERR_ARG;
    e->P = params; e->G = grads; e->SH = (char*)shadow; e->ws = (char*)workspace;
    e->grads_zero = false;                 // a newly bound gradient buffer: nothing is known about its contents
    e->ws_zeroed = false; e->padT = -1;
    e->drop_graphs();                         // captured against the old buffers
    char* mws = e->ws + e->ws_mag;
    e->pkw = {e->P + e->mag_whv, e->P + e->mag_wha, e->P + e->mag_wv, e->P + e->mag_wa, mws + e->mw.We, mws + e->mw.Wv, mws + e->mw.Wa,
              MagDims{0, e->c.hidden_size, e->c.visual_dim, e->c.acoustic_dim, e->mw.Vp, e->mw.Ap}, e->c.dtype};
    return MB_OK;
}

int mb_bert_sync_weights(mb_bert_engine* e, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (!e->P) return MB_ERR_ARG;
    if (e->c.dtype == DT_BF16)
        CK(convert(DT_BF16, e->P + e->sh_begin, e->SH + e->sh_begin * 2, e->sh_end - e->sh_begin, st));
    const mb_bert_config& c = e->c;
    MagDims d = {0, c.hidden_size, c.visual_dim, c.acoustic_dim, e->mw.Vp, e->mw.Ap};
    char* mws = e->ws + e->ws_mag;
    CK(mag_pack_weights(c.dtype, e->P + e->mag_whv, e->P + e->mag_wha, e->P + e->mag_wv, e->P + e->mag_wa, mws + e->mw.We,
                        mws + e->mw.Wv, mws + e->mw.Wa, d, st));
    return MB_OK;
}

// The forward of layers [l0, l1): `first` = with the pass set-up, the embeddings and MAG in front; `last` = with the pooler, the classifier
// and the loss behind.  mb_bert_forward is the whole range; the sharded data-parallel step runs it in pieces whose weights arrive by
// all-gather (csrc/comm.h: cut mode).
static int bert_forward_range(mb_bert_engine* e, const int64_t* input_ids, const float* visual, const float* acoustic,
                              const int64_t* attention_mask, const int64_t* token_type_ids, const float* labels, int B, int L,
                              int training, uint64_t seed, uint64_t step, float* logits, float* loss, float* loss_run, void* stream,
                              int l0, int l1, bool first, bool last) {
    hipStream_t st = (hipStream_t)stream;
    const mb_bert_config& c = e->c;
    if (!e->P || !e->ws) return MB_ERR_ARG;
    if (B < 1 || B > c.max_batch || L < 1 || L > c.max_seq) return MB_ERR_SHAPE;
    if ((!input_ids && !e->emb_in) || !visual || !acoustic || !attention_mask || !token_type_ids || !logits) return MB_ERR_ARG;
    const int dt = c.dtype, H = c.hidden_size, I = c.intermediate_size, T = B * L, nh = c.num_heads;
    if (e->emb_in) input_ids = nullptr;           // inputs_embeds given: the word-table gather (and its gradient scatter) is skipped
    e->ids = input_ids; e->seg = token_type_ids; e->mask = attention_mask; e->ran_forward = true;
    e->B = B; e->L = L; e->training = training; e->seed = seed; e->step = step; e->logits = logits;
    float* P = e->P;
    char* ws = e->ws;
    if (first && !e->capturing) CK(prepare_pass(e, T, st));
    // embeddings (bert.py:211-216)
    if (first) CK(embed_ln_forward(dt, input_ids, token_type_ids, e->emb_in ? e->emb_in : P + e->word, P + e->pos, P + e->type, P + e->emb_lnw, P + e->emb_lnb,
                        c.layer_norm_eps, ws + e->ws_emb, (float*)(ws + e->ws_emb_st), (float*)(ws + e->ws_emb_st) + T, B, L,
                        H, e->key(SITE_EMB, c.hidden_dropout), st, e->pos_ids));
    // MAG (bert.py:219): packed weights are refreshed every pass (5.5 MB) so optimizer steps are seen -- by a launch here, or, in the
    // single-call step, by extra blocks of the step prologue
    if (first) CK(mag_fwd_impl(dt, ws + e->ws_emb, visual, acoustic, P + e->mag_whv, P + e->mag_bhv, P + e->mag_wha, P + e->mag_bha,
                    P + e->mag_wv, P + e->mag_bv, P + e->mag_wa, P + e->mag_ba, P + e->mag_lnw, P + e->mag_lnb,
                    c.mag_layer_norm_eps, c.beta_shift, e->key(SITE_MAG, c.mag_dropout), ws + e->ws_x[0], ws + e->ws_mag,
                    e->mw, T, H, c.visual_dim, c.acoustic_dim, !(e->in_step && e->packed_w), st, true, e->in_step && e->packed));
    // encoder (bert.py:221-229)
    for (int l = l0; l < l1; ++l) {
        const LayerOff& o = e->lo[l];
        const LayerWs& w = e->lw[l];
        const char* x = ws + e->ws_x[l];
        CK(gemm(dt, GEMM_NT, EPI_BIAS, T, 3 * H, H, x, H, e->W(o.wqkv), H, ws + w.qkv, 3 * H, nullptr, nullptr, P + o.bqkv,
                nullptr, 0, kNoDrop, 1, 0, st));
        CK(attention_forward(dt, ws + w.qkv, attention_mask, ws + w.ctx, B, L, nh,
                             e->key(SITE_LAYER0 + 4 * l + 0, c.attn_dropout), st,
                             e->attn_out ? e->attn_out + (size_t)l * B * nh * L * L : nullptr,
                             e->head_mask ? e->head_mask + (size_t)l * nh : nullptr));
        CK(gemm(dt, GEMM_NT, EPI_BIAS_DROP_RES, T, H, H, ws + w.ctx, H, e->W(o.wo), H, ws + w.s1, H, nullptr, nullptr,
                P + o.bo, x, H, e->key(SITE_LAYER0 + 4 * l + 1, c.hidden_dropout), 1, 0, st));
        // (the two LayerNorm launches of a layer touch the weights of the GEMMs behind them: W1 | W2, then the next layer's Wqkv | Wo)
        const size_t wes = dt == DT_BF16 ? 2 : 4;
        const Prefetch pf1 = {e->prefetch ? e->W(o.w1) : nullptr, (size_t)2 * I * H * wes, nullptr};
        // (not across the seam of a forward cut into pieces: the next piece's weights may still be on their way -- sharded update)
        const Prefetch pf2 = {(e->prefetch && l + 1 < c.num_layers && (l + 1 < l1 || last)) ? e->W(e->lo[l + 1].wqkv) : nullptr, (size_t)4 * H * H * wes, nullptr};
        CK(ln_forward(dt, ws + w.s1, P + o.ln1w, P + o.ln1b, c.layer_norm_eps, ws + w.y1, (float*)(ws + w.st1),
                      (float*)(ws + w.st1) + T, T, H, kNoDrop, st, pf1));
        CK(gemm(dt, GEMM_NT, EPI_BIAS_GELU, T, I, H, ws + w.y1, H, e->W(o.w1), H, ws + w.u, I, ws + w.g, nullptr, P + o.b1,
                nullptr, 0, kNoDrop, 1, 0, st));
        CK(gemm(dt, GEMM_NT, EPI_BIAS_DROP_RES, T, H, I, ws + w.g, I, e->W(o.w2), I, ws + w.s2, H, nullptr, nullptr,
                P + o.b2, ws + w.y1, H, e->key(SITE_LAYER0 + 4 * l + 2, c.hidden_dropout), 1, 0, st));
        CK(ln_forward(dt, ws + w.s2, P + o.ln2w, P + o.ln2b, c.layer_norm_eps, ws + e->ws_x[l + 1], (float*)(ws + w.st2),
                      (float*)(ws + w.st2) + T, T, H, kNoDrop, st, pf2));
    }
    if (!last) return MB_OK;
    // pooler + classifier (+ MSE) (bert.py:231, 304-307; multimodal_driver.py:372-373)
    float* z = (float*)(ws + e->ws_head_z);
    CK(gemm(dt, GEMM_NT, EPI_BIAS_F32, B, H, H, ws + e->ws_x[c.num_layers], L * H, e->W(e->wp), H, nullptr, H, nullptr, z,
            P + e->bp, nullptr, 0, kNoDrop, 1, 64, st));
    if (loss && !(e->in_step && e->loss_cleared)) CK(zero_fill(loss, 4, st));
    CK(head_forward(z, P + e->wc, P + e->bc, labels, (float*)(ws + e->ws_head_pooled), logits, loss, loss_run, B, H,
                    c.num_labels, e->key(SITE_HEAD, c.hidden_dropout), st));
    return MB_OK;
}

int mb_bert_forward(mb_bert_engine* e, const int64_t* input_ids, const float* visual, const float* acoustic,
                    const int64_t* attention_mask, const int64_t* token_type_ids, const float* labels, int B, int L,
                    int training, uint64_t seed, uint64_t step, float* logits, float* loss, float* loss_run, void* stream) {
    return bert_forward_range(e, input_ids, visual, acoustic, attention_mask, token_type_ids, labels, B, L, training, seed, step, logits,
                              loss, loss_run, stream, 0, e->c.num_layers, true, true);
}

int mb_bert_backward(mb_bert_engine* e, const float* dlogits, const float* labels, float loss_scale, int stage_begin,
                     int stage_end, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const mb_bert_config& c = e->c;
    if (!e->G || !e->ran_forward) return MB_ERR_ARG;
    const int dt = c.dtype, H = c.hidden_size, I = c.intermediate_size, B = e->B, L = e->L, T = B * L, nh = c.num_heads;
    const int NL = c.num_layers;
    const int Tk = (int)align_up((size_t)T, 64);      // zero-padded reduction length of the wgrad GEMMs
    // Riders (kernels.h AdamRide, enqueue_step): up to `budget` parameters from the TOP of what is final while layer l's backward runs --
    // the GEMM weights of layers l+1 .. NL-1 minus what earlier launches of this backward took -- as `blocks` extra workgroups of a
    // launch.  The sweep at the end of the step is then ONE range [0, ride_cursor) + the rest of the slabs.
    auto take_ride = [&](int l, size_t budget, int blocks) -> AdamRide {
        AdamRide r = {};
        if (!e->ride_m || !e->ride_v || l + 1 >= e->c.num_layers || blocks < 8 || e->ride_cursor <= e->lo[l + 1].wqkv) return r;
        size_t take = std::min(e->ride_cursor - e->lo[l + 1].wqkv, budget) / 1024 * 1024;
        const size_t re = e->ride_cursor, rb = re - take;
        const bool sh_ok = e->c.dtype != DT_BF16 || (e->sh_begin <= rb && re <= e->sh_end);
        if (take == 0 || rb % 4 || !sh_ok) return r;
        const bool keep = e->keep_in_step() && e->stale_begin <= rb && re <= e->stale_end;
        r = AdamRide{e->P + rb, e->G + rb, e->ride_m + rb, e->ride_v + rb, e->c.dtype == DT_BF16 ? (bf16*)(e->SH + rb * 2) : nullptr,
                     take / 4, e->adam_state(e->ws), blocks / 8 * 8, keep ? 0 : 1};
        e->ride_cursor = rb;
        return r;
    };
    // (Riders in the LayerNorm-backward launches -- 150 latency-bound workgroups on 256 CUs, 24 launches per step -- were built and
    //  measured: no gain at any budget, +0.09 ms when they replace the weight-gradient riders: profiles/r06_adamw_ride_ln.txt.  What
    //  pays is a CU that has NOTHING else to do for tens of microseconds.)
    if (stage_begin < 0) stage_begin = 0;
    if (stage_end > NL + 2) stage_end = NL + 2;
    if (stage_begin == 0) CK(e->begin_backward_pass(e->G, st));
    float* P = e->P; float* G = e->G;
    char* ws = e->ws;
    const bool hd = e->training && c.hidden_dropout > 0.f;
    const GradAcc acc = e->acc_of(ws, G);          // deterministic mode: where the multi-writer column sums go
    for (int stage = stage_begin; stage < stage_end; ++stage) {
        if (stage == 0) {
            // ---- head + pooler
            // (one launch: dz, the classifier gradients, the pooler bias gradient = column sums of dz, and the clearing of the token
            //  gradient buffer whose [CLS] rows the dgrad below fills)
            CK(head_backward(dt, dlogits, e->logits, labels, loss_scale, (const float*)(ws + e->ws_head_pooled), P + e->wc,
                             ws + e->ws_dz, G + e->wc, G + e->bc, B, H, c.num_labels, e->key(SITE_HEAD, c.hidden_dropout),
                             st, acc, G + e->bp, ws + e->ws_dxa, (size_t)T * H * esize(dt)));
            const char* xf = ws + e->ws_x[NL];
            CK(gemm(dt, GEMM_TN, EPI_ACCUM_F32, H, H, B, ws + e->ws_dz, H, xf, L * H, nullptr, H, nullptr, G + e->wp, nullptr,
                    nullptr, 0, kNoDrop, 1, 64, st));
            CK(gemm(dt, GEMM_NN, EPI_ADD_RES, B, H, H, ws + e->ws_dz, H, e->W(e->wp), H, ws + e->ws_dxa, L * H, nullptr,
                    nullptr, nullptr, nullptr, 0, kNoDrop, 1, 64, st));
        } else if (stage <= NL) {
            const int l = NL - stage;
            const LayerOff& o = e->lo[l];
            const LayerWs& w = e->lw[l];
            char* dx = ws + e->ws_dxa;     // grad wrt x[l+1] on entry, wrt x[l] on exit
            char* dy1 = ws + e->ws_dxb;
            const int par = l & 1;                      // the grouped wgrad of layer l reads these while layer l-1 runs
            char* dsA = ws + e->ws_ds[par];             // LN2 (FFN output) backward
            char* dzdA = hd ? ws + e->ws_dzd[par] : dsA;
            char* dsB = ws + e->ws_ds2[par];            // LN1 (attention output) backward
            char* dzdB = hd ? ws + e->ws_dzd2[par] : dsB;
            char* du = ws + e->ws_du[par];
            char* dqkv = ws + e->ws_dqkv[par];
            // wgrad GEMMs go to the side stream: they only read (dY, saved X) and accumulate into G, so they overlap the
            // dgrad chain; dY buffers are per-LayerNorm (A/B) and the stage ends with a join, which keeps them race-free.
            hipStream_t ss = st;
            if (e->overlap_wgrad) {
                CK(ensure_side(e));
                ss = e->side;
            }
            hipEvent_t* sev = e->overlap_wgrad ? &e->evs[(size_t)l * 5] : nullptr;
            auto fork = [&](int k) -> int {      // side stream may start once main has produced dY number k
                if (ss == st) return 0;
                int r = (int)hipEventRecord(sev[k], st);
                if (r) return r;
                return (int)hipStreamWaitEvent(ss, sev[k], 0);
            };
            int nblk = 0;
            float* lnp_a = (float*)(ws + e->ws_lnp_a) + (size_t)l * e->lnp_stride;
            float* lnp_b = (float*)(ws + e->ws_lnp_b) + (size_t)l * e->lnp_stride;
            // single-call step: nobody needs this layer's LayerNorm / bias gradients before AdamW -> all layers reduced at once
            const bool defer_ln = e->in_step && !e->stage_mode && NL + 2 <= MB_LN_MAX_LAYERS;      // (reduced in the last stage)
            // LN2 + dropout backward (column sums -> per-block partial slabs, reduced once per layer below)
            CK(ln_backward_partials(dt, dx, ws + w.s2, P + o.ln2w, (const float*)(ws + w.st2), (const float*)(ws + w.st2) + T, dsA,
                                    hd ? dzdA : nullptr, lnp_a, &nblk, T, H, e->key(SITE_LAYER0 + 4 * l + 2, c.hidden_dropout), st,
                                    // ... and touches W1 | W2 for the two FFN dgrads behind it (common.h Prefetch)
                                    Prefetch{e->prefetch ? e->W(o.w1) : nullptr, (size_t)2 * I * H * (dt == DT_BF16 ? 2 : 4), nullptr}));
            // The four weight gradients of the layer: one grouped launch once dqkv exists (MB_GROUP_WGRAD=0: four launches,
            // each forked as soon as its dY is final).
            GemmArgs wg[4] = {wgrad_args(H, I, Tk, dzdA, H, ws + w.g, I, G + o.w2, I),
                              wgrad_args(I, H, Tk, du, I, ws + w.y1, H, G + o.w1, H),
                              wgrad_args(H, H, Tk, dzdB, H, ws + w.ctx, H, G + o.wo, H),
                              wgrad_args(3 * H, H, Tk, dqkv, 3 * H, ws + e->ws_x[l], H, G + o.wqkv, H)};
            const bool grouped = e->grouped;
            if (grouped)
                for (GemmArgs& a : wg) a.overwrite = e->ow_pass ? 1 : 0;       // whole-tile, no split-K launches only
            const bool inl = grouped && !e->deferred;         // grouped launch in line on the caller's stream (no overlap)
            // dX = dY . W + R (GEMM_NN, EPI_ADD_RES), with riders when the launch is the 64 x 64 three-slot kernel and leaves block slots free
            // (Touching the forward activations the grouped weight gradient multiplies with -- g, y1, x_l, 22 MB from HBM -- makes that launch
            //  1.2 .. 3.2 us faster, but every carrier tried pays more than that: separate launches 3 x 4.7 us, the attention backward +1.8 ..
            //  2.2 us (register or LDS-DMA loads), these riders +3.3 us per dgrad launch: profiles/r06_wgrad_operand_touch.txt)
            // (mode EPI_DGELU: the ffn2 dgrad with the fused bias gradient `colsum`; 56 of its CUs hold one tile instead of two)
            auto dgrad_ride = [&](int mode, int Mo, int No, int Ko, const void* dY, int ldy, const void* Wt, int ldw, void* dX, int ldx, const void* R, int ldr,
                                  float* colsum) -> int {
                const RideOpts ro = {e->ride_dgrad, e->ride_dgrad_blocks, e->ride_dgrad_params, e->ride_dgelu_params};
                return dgrad_with_riders(dt, mode, Mo, No, Ko, dY, ldy, Wt, ldw, dX, ldx, R, ldr, colsum, kNoDrop, acc, st, inl && e->ride_m != nullptr, ro,
                                         e->cu_count(), [&](size_t budget, int blocks) { return take_ride(l, budget, blocks); });
            };
            int wtile = e->group_wgrad;
            if (grouped && wtile == 256 && !gemm_grouped_tn_ok(dt, wg, 4, 256)) wtile = 128;      // (fewer than three k stages: the 128 x 128 kernel)
            if (grouped && !gemm_grouped_tn_ok(dt, wg, 4, wtile)) return MB_ERR_SHAPE;
            if (!grouped) {
            CK(fork(0));
            CK(wgrad(dt, H, I, Tk, dzdA, H, ws + w.g, I, G + o.w2, I, ss));
            }
            // du = (dzd . W2) * gelu'(u), with the intermediate bias gradient (column sums of du) fused into the epilogue
            CK(dgrad_ride(EPI_DGELU, T, I, H, dzdA, H, e->W(o.w2), I, du, I, ws + w.u, I, G + o.b1));
            if (!grouped) {
            CK(fork(1));
            CK(wgrad(dt, I, H, Tk, du, I, ws + w.y1, H, G + o.w1, H, ss));
            }
            CK(dgrad_ride(EPI_ADD_RES, T, H, I, du, I, e->W(o.w1), H, dy1, H, dsA, H, nullptr));
            // LN1 + dropout backward
            Prefetch pf_attn = {e->prefetch ? e->W(o.wqkv) : nullptr, (size_t)4 * H * H * (dt == DT_BF16 ? 2 : 4), nullptr};
            if (e->prefetch && e->pf_qkv > 0) {
                const size_t es = dt == DT_BF16 ? 2 : 4, qb = (size_t)T * 3 * H * es;
                pf_attn.p2 = ws + w.qkv;
                pf_attn.bytes2 = (w.ctx - w.qkv) - qb <= 4096 ? (w.ctx - w.qkv) + (size_t)T * H * es : qb;      // context rows follow unless T < max
                pf_attn.stride2 = (uint32_t)e->pf_qkv;
            }
            CK(ln_backward_partials(dt, dy1, ws + w.s1, P + o.ln1w, (const float*)(ws + w.st1), (const float*)(ws + w.st1) + T, dsB,
                                    hd ? dzdB : nullptr, lnp_b, &nblk, T, H, e->key(SITE_LAYER0 + 4 * l + 1, c.hidden_dropout), st,
                                    // ... Wqkv | Wo for the attention-side dgrads, and with the loads that leaves over the q | k | v (+ context)
                                    // rows this layer saved in the forward, which the attention backward two launches on finds in HBM
                                    pf_attn));
            if (!grouped) {
            CK(fork(2));
            CK(wgrad(dt, H, H, Tk, dzdB, H, ws + w.ctx, H, G + o.wo, H, ss));
            }
            e->lnp_nblk = nblk;
            if (!defer_ln) {
                float* const dst6[6] = {G + o.ln2w, G + o.ln2b, G + o.b2, G + o.ln1w, G + o.ln1b, G + o.bo};
                CK(ln_reduce_partials(lnp_a, lnp_b, nblk, H, dst6, st, acc));
            }
            CK(gemm(dt, GEMM_NN, EPI_ADD_RES, T, H, H, dzdB, H, e->W(o.wo), H, ws + e->ws_dctx, H, nullptr, nullptr, nullptr,
                    nullptr, 0, kNoDrop, 1, 0, st));
            // attention backward; the fused-QKV bias gradient (column sums of dqkv) is accumulated inside the kernel
            // (riders: the launch's empty block slots -- L <= 64: 448 of 1024 next to a latency-bound kernel; L = 128: the 128 CUs its second
            //  round leaves idle -- carry a piece of the optimizer update; MB_ADAMW_RIDE_ATTN=0 turns them off)
            AdamRide ra = {};
            if (e->ride_attn && inl && e->ride_m) {
                int free_slots = attention_backward_free_slots(dt, L, B * nh, e->cu_count());
                if (e->ride_attn_blocks > 0) free_slots = e->ride_attn_blocks;
                const int blocks = std::min(free_slots, 2 * e->cu_count()) / 8 * 8;
                if (blocks >= 8) {
                    // L = 128: 128 whole CUs for about half the launch (~41 GB/s each): 20,480 parameters per rider workgroup -- same box, B = 32:
                    // 4.743 ms without | 4.68 at 1.5 M | 4.652 at 2.6 M (this) | 4.66 at 3.5 M | 4.74 at 5 M per launch.  L <= 64: shared CUs, but
                    // the kernel is latency-bound and barely notices: 3.320 ms without | 3.289 at 1.5 M | 3.263 at 2 M | 3.257 at 2.5 M (this:
                    // 1,040 per token) | 3.270 at 3 M (profiles/r06_adamw_ride_attn.txt).  Later in the round the dgrad launches around it moved to
                    // the 128 x 64 ping-pong tile, whose 16 idle CUs carry 0.2 - 0.26 M where the 64 x 64 kernel's free slots carried 1.25 M:
                    // with the sweep that much longer the optimum moved up -- 3.169 ms at 2.5 M | 3.157 at 3 M | 3.152 at 3.5 M | 3.140 at 4 M
                    // (this: 1,650 per token) | 3.140 at 4.5 M (profiles/r06_ride_budget3.txt)
                    const size_t budget = e->ride_attn_params > 0 ? (size_t)e->ride_attn_params
                                                                  : (L > 64 ? (size_t)blocks * 20480 : (size_t)1650 * (size_t)T);
                    ra = take_ride(l, budget / 1024 * 1024, blocks);
                }
            }
            CK(attention_backward(dt, ws + w.qkv, e->mask, ws + w.ctx, ws + e->ws_dctx, dqkv, G + o.bqkv, B, L, nh,
                                  e->key(SITE_LAYER0 + 4 * l + 0, c.attn_dropout), st,
                                  e->head_mask ? e->head_mask + (size_t)l * nh : nullptr, acc, ra.blocks ? &ra : nullptr));
            // (experiment) the update inside the launch: the tile of the gradient becomes the new parameters -- so every reader of the
            // OLD weights of this layer (the qkv dgrad below) goes first
            const bool fuse = inl && e->fuse_m && e->fuse_v && e->ow_pass && e->group_wgrad == 128;
            if (fuse) {
                const size_t offs[4] = {o.w2, o.w1, o.wo, o.wqkv};
                for (int k = 0; k < 4; ++k) {
                    wg[k].C = P + offs[k]; wg[k].C2 = e->fuse_m + offs[k]; wg[k].R = e->fuse_v + offs[k];
                    wg[k].bias = (const float*)e->adam_state(ws);
                    wg[k].colsum = dt == DT_BF16 ? (float*)(e->SH + offs[k] * 2) : nullptr;
                }
                CK(gemm(dt, GEMM_NN, EPI_ADD_RES, T, H, 3 * H, dqkv, 3 * H, e->W(o.wqkv), H, dx, H, nullptr, nullptr,
                        nullptr, dsB, H, kNoDrop, 1, 0, st));
            }
            // riders: an optimizer update in the empty slots of this launch (take_ride, below the loop header)
            AdamRide ride = {};
            if (inl && !fuse) {
                int tiles = 0;
                const int bm = wtile == 256 ? 256 : wtile, bn = wtile == 256 ? 128 : wtile;
                for (const GemmArgs& a : wg) tiles += (a.M / bm) * (a.N / bn);
                const int slots = e->cu_count() * (wtile == 256 ? 1 : wtile == 128 ? 2 : 0);
                // what 40 CUs stream while the tiles multiply (~41 GB/s per CU; same-box sweeps in profiles/r06_adamw_ride_budget.txt,
                // r06_adamw_ride_ab.txt: 2.5 M parameters at T = 2400, 3.5 M at T = 4096; more stretches the launch)
                const size_t budget = e->ride_params > 0 ? (size_t)e->ride_params : std::min((size_t)1000 * (size_t)Tk, (size_t)1100000 + (size_t)590 * (size_t)Tk);
                ride = take_ride(l, budget, e->ride_blocks > 0 ? e->ride_blocks : (slots - tiles) / 8 * 8);
            }
            auto launch_group = [&]() -> int {
                if (inl) {
                    if (e->prof) CK((int)hipEventRecord(e->pev[2 * l], st));
                    CK(gemm_grouped_tn_launch(dt, wg, 4, wtile, st, 0, fuse, ride.blocks ? &ride : nullptr));
                    if (e->prof) CK((int)hipEventRecord(e->pev[2 * l + 1], st));
                    return MB_OK;
                }
                CK(fork(3));
                if (e->prof) CK((int)hipEventRecord(e->pev[2 * l], ss));
                CK(gemm_grouped_tn_launch(dt, wg, 4, wtile, ss));
                if (e->prof) CK((int)hipEventRecord(e->pev[2 * l + 1], ss));
                return (int)hipEventRecord(sev[4], ss);       // "weight gradients (or updated weights) of layer l are final"
            };
            if (grouped) CK(launch_group());
            if (!grouped) {
                CK(fork(3));
                CK(wgrad(dt, 3 * H, H, Tk, dqkv, 3 * H, ws + e->ws_x[l], H, G + o.wqkv, H, ss));
            }
            if (!fuse) CK(dgrad_ride(EPI_ADD_RES, T, H, 3 * H, dqkv, 3 * H, e->W(o.wqkv), H, dx, H, dsB, H, nullptr));
            if (ss != st && !grouped) {      // join: the stage's gradients are complete (and dY buffers reusable) once main passes this
                CK((int)hipEventRecord(sev[4], ss));
                CK((int)hipStreamWaitEvent(st, sev[4], 0));
            }
            // deferred join: the grouped wgrad of layer l keeps running under the dgrad chain of layer l-1; main only
            // waits for layer l+1's (whose dY buffers, same parity as l-1, are written next).  What is final on `st` when
            // this stage returns is therefore: weights of layer l+1, biases / LayerNorm of layer l (mb_bert_stage_grad_ranges)
            if (e->deferred && l + 1 < NL) CK((int)hipStreamWaitEvent(st, e->evs[(size_t)(l + 1) * 5 + 4], 0));
        } else {
            // ---- MAG + embeddings
            if (e->deferred && e->side) CK((int)hipStreamWaitEvent(st, e->evs[4], 0));      // weight gradients of layer 0
            char* dx = ws + e->ws_dxa;
            char* de = ws + e->ws_dxb;
            // single-call step: the six column sums of MAG's gate go to partial slabs like the LayerNorm ones, and ONE launch reduces
            // every layer's and MAG's slabs (nobody needs these small gradients before AdamW)
            const bool defer_ln = e->in_step && !e->stage_mode && NL + 2 <= MB_LN_MAX_LAYERS && NL > 0;
            float* mpa = (float*)(ws + e->ws_lnp_a) + (size_t)NL * e->lnp_stride;
            float* mpb = (float*)(ws + e->ws_lnp_b) + (size_t)NL * e->lnp_stride;
            int mblk = 0;
            CK(mag_bwd_impl(dt, dx, ws + e->ws_emb, P + e->mag_bhv, P + e->mag_bha, P + e->mag_bv, P + e->mag_ba,
                            P + e->mag_lnw, c.beta_shift, e->key(SITE_MAG, c.mag_dropout), ws + e->ws_mag, e->mw, de, nullptr,
                            nullptr, G + e->mag_whv, G + e->mag_bhv, G + e->mag_wha, G + e->mag_bha, G + e->mag_wv,
                            G + e->mag_bv, G + e->mag_wa, G + e->mag_ba, G + e->mag_lnw, G + e->mag_lnb, T, H, c.visual_dim,
                            c.acoustic_dim, true, st, acc, true, mpa, mpb, &mblk, e->ow_pass));
            int eblk = 0;
            CK(embed_ln_backward(dt, de, e->ids, e->seg, e->ids ? P + e->word : e->emb_in, P + e->pos, P + e->type, P + e->emb_lnw,
                                 (const float*)(ws + e->ws_emb_st), (const float*)(ws + e->ws_emb_st) + T,
                                 (float*)(ws + e->ws_dsum), e->ids ? G + e->word : nullptr, G + e->pos, G + e->type, G + e->emb_lnw,
                                 G + e->emb_lnb, B, L, H, c.pad_token_id, e->key(SITE_EMB, c.hidden_dropout), st, e->pos_ids, acc,
                                 (float*)(ws + e->ws_lnp_a) + (size_t)(NL + 1) * e->lnp_stride, &eblk,
                                 (float*)(ws + e->ws_lnp_b) + (size_t)(NL + 1) * e->lnp_stride,
                                 (e->in_step && e->counted && e->ids && !e->emb_in) ? (int*)(ws + e->idcnt_off) : nullptr));
            {
                // ONE reduction launch: every layer's LayerNorm / bias slabs (single-call step only: the other modes reduced them per
                // layer, their stage hooks need them early), MAG's six sums (slot NL) and the embedding LayerNorm's two (slot NL + 1).
                // Same slabs and summation order in every mode (deterministic mode: bit-identical trajectories across the modes).
                const int first = defer_ln ? 0 : NL;
                LnReduceDst dst = {};
                for (int k = first; k < NL; ++k) {
                    const LayerOff& ok = e->lo[k];
                    float* const d6[6] = {G + ok.ln2w, G + ok.ln2b, G + ok.b2, G + ok.ln1w, G + ok.ln1b, G + ok.bo};
                    for (int q = 0; q < 6; ++q) dst.d[k - first][q] = d6[q];
                }
                float* const m6[6] = {G + e->mag_bhv, G + e->mag_bha, G + e->mag_bv, G + e->mag_ba, G + e->mag_lnw, G + e->mag_lnb};
                for (int q = 0; q < 6; ++q) dst.d[NL - first][q] = m6[q];
                dst.nblk[NL - first] = mblk; dst.nblk[NL + 1 - first] = eblk;         // (their kernels: 8 rows per block; ln_bwd: 16)
                dst.d[NL + 1 - first][0] = G + e->emb_lnw; dst.d[NL + 1 - first][1] = G + e->emb_lnb;   // (rows 0 / 1 of set a)
                if (e->pos_ids == nullptr) {       // the one-launch embedding backward: token-type sums in row 2 of set a / row 0 of set b
                    dst.d[NL + 1 - first][2] = G + e->type; dst.d[NL + 1 - first][3] = G + e->type + H;
                }
                CK(ln_reduce_partials_layers((const float*)(ws + e->ws_lnp_a) + (size_t)first * e->lnp_stride,
                                             (const float*)(ws + e->ws_lnp_b) + (size_t)first * e->lnp_stride, e->lnp_stride, NL + 2 - first,
                                             e->lnp_nblk, H, dst, st, acc));
            }
            // deterministic mode: the integer sums become part of the fp32 gradients before anybody (AdamW, an exchange) reads them
            CK(grad_fold(acc, G, e->det_begin, e->det_end, st));
        }
    }
    return MB_OK;
}

// ------------------------------------------------------------------------------------------------ whole step
// One optimizer step of train_epoch (/root/reference/multimodal_driver.py:354-388: batch -> forward -> MSE -> backward ->
// optimizer.step() -> optimizer.zero_grad()) as two launches: the step prologue (this step's batch, dropout keys and AdamW
// scalars into device memory) and a replayed hipGraph holding every other kernel of the step -- ONE in-order kernel sequence
// (the grouped weight-gradient launches run in line: a graph with a side-stream fork / join replays on a slow path).
// mode 1 = graph replay (captured on first use per shape), mode 2 = the same kernel sequence launched one by one (A/B
// reference for the graph; also what runs while profiling events are on).  The prologue also converts the modality tensors into
// MAG's packed GEMM operands, counts the occurrences of every token id and clears the loss accumulator (rowops.hip).

// AdamW of flat range [b, en) of the decay slab (GEMM weights), inside a step (scalars from device memory)
static int adamw_decay_range(mb_bert_engine* e, float* m, float* v, size_t b, size_t en, hipStream_t st) {
    if (en <= b) return MB_OK;
    const AdamArgs none = {};
    const bool keep = e->keep_in_step();          // the layers' GEMM weight gradients: overwritten by the next backward, not zeroed
    auto clampr = [&](size_t x) { return x < b ? (size_t)0 : (x > en ? en - b : x - b); };
    const size_t shb = clampr(e->sh_begin), she = clampr(e->sh_end);
    const size_t kb = keep ? clampr(e->stale_begin) : 0, ke = keep ? clampr(e->stale_end) : 0;
    void* sh = e->c.dtype == DT_BF16 ? (void*)(e->SH + b * 2) : nullptr;
    return adamw_step(e->P + b, e->G + b, m + b, v + b, sh, en - b, en - b, shb, she, none, 1, st, e->adam_state(e->ws), kb, ke);
}

// The step as `nseg` segments (train_step_impl).  nseg == 1: forward, backward, optimizer.  MB_ADAMW_OVERLAP=C (C layers per chunk,
// nseg = layers / C + 1): segment i ends with the backward of a chunk of C layers; the host then forks the AdamW of THAT chunk's
// GEMM weights (7.08 M parameters per layer, 77 % of the model: HBM-bound) onto the optimizer side stream, where it runs under the
// MFMA-bound backward of the layers below; the last segment holds the MAG / embedding backward and the optimizer of everything
// else, and the step ends with the join.  Nothing later in the step reads a finished layer's weights, gradients or shadow.
static int enqueue_step(mb_bert_engine* e, int seg, int nseg, int B, int L, float* logits, float* loss, float* loss_run, float* m, float* v,
                        float loss_scale, hipStream_t st) {
    char* ws = e->ws;
    const int NL = e->c.num_layers;
    const float* lab = (const float*)(ws + e->ws_in_lab);
    float* keep_attn = e->attn_out;
    e->attn_out = nullptr;                      // optional outputs belong to explicit forwards, never to a (captured) training step
    struct Restore { mb_bert_engine* e; float* p; ~Restore() { e->attn_out = p; } } restore{e, keep_attn};
    if (e->head_mask || e->emb_in || e->pos_ids) return MB_ERR_MODE;     // head_mask / inputs_embeds / position_ids are arguments of explicit forwards only
    const int C = nseg > 1 ? NL / (nseg - 1) : 0;
    if (seg == 0)
        CK(mb_bert_forward(e, (const int64_t*)(ws + e->ws_in_ids), (const float*)(ws + e->ws_in_vis), (const float*)(ws + e->ws_in_aco),
                           (const int64_t*)(ws + e->ws_in_mask), (const int64_t*)(ws + e->ws_in_seg), lab, B, L, 1, 0, 0, logits, loss,
                           loss_run, st));
    // backward stages: 0 = head, 1 .. NL = layers NL-1 .. 0, NL+1 = MAG + embeddings
    const int sb = nseg == 1 ? 0 : (seg == 0 ? 0 : 1 + seg * C);
    const int se = nseg == 1 ? NL + 2 : (seg + 1 < nseg ? 1 + (seg + 1) * C : NL + 2);
    // (experiment) the layers' weights are updated by their own weight-gradient launches: single segment, known-zero gradients,
    // in-line 128 x 128 grouped launches, the layers' GEMM weights at the head of the decay slab
    const bool fuse = e->adam_in_wgrad && m && v && nseg == 1 && e->ow_pass && e->grouped && !e->deferred && e->group_wgrad == 128 && NL > 0 &&
                      e->lo[0].wqkv == 0 && !e->prof;
    e->fuse_m = fuse ? m : nullptr; e->fuse_v = fuse ? v : nullptr;
    // riders (MB_ADAMW_RIDE): layers 1 .. NL-1 are updated inside the weight-gradient launches of layers 0 .. NL-2; whether a launch
    // really carried one is decided there, so the sweep below asks the engine which layers are still to do
    const bool ride = e->adam_ride && !fuse && m && v && nseg == 1 && e->grouped && !e->deferred && NL > 1 && e->lo[0].wqkv == 0 && !e->prof &&
                      (e->group_wgrad == 128 || e->group_wgrad == 256);
    e->ride_m = ride ? m : nullptr; e->ride_v = ride ? v : nullptr;
    e->ride_cursor = e->wp;
    const int rb = mb_bert_backward(e, nullptr, lab, loss_scale, sb, se, st);
    e->fuse_m = e->fuse_v = nullptr;
    e->ride_m = e->ride_v = nullptr;
    CK(rb);
    if (m && v && seg == nseg - 1) {
        const AdamArgs none = {};
        const size_t nd = e->n_decay, n = e->n_params;
        CK(e->prof_mark(2 * NL, st));
        // (with chunks on the side stream, what is left of the decay slab: the pooler weight .. the classifier weight)
        if (ride) CK(adamw_decay_range(e, m, v, 0, e->ride_cursor, st));       // what no launch carried (layer 0 always)
        CK(adamw_decay_range(e, m, v, (nseg == 1 && !fuse && !ride) ? 0 : e->wp, nd, st));
        CK(adamw_step(e->P + nd, e->G + nd, m + nd, v + nd, nullptr, n - nd, 0, 0, 0, none, 1, st, e->adam_state(ws) + 1));
        CK(e->prof_mark(2 * NL + 1, st));
    }
    return MB_OK;
}

// host side of MB_ADAMW_OVERLAP, after segment `seg` was enqueued (never captured): fork the chunk's optimizer / join at the end
static int between_segments(mb_bert_engine* e, int seg, int nseg, float* m, float* v, hipStream_t st) {
    if (nseg == 1 || !m || !v) return MB_OK;
    const int NL = e->c.num_layers, C = NL / (nseg - 1);
    if (seg + 1 < nseg) {
        const int l_lo = NL - (seg + 1) * C, l_hi = NL - seg * C;       // layers [l_lo, l_hi) finished in this segment
        CK((int)hipEventRecord(e->opt_ev[seg], st));
        CK((int)hipStreamWaitEvent(e->opt_side, e->opt_ev[seg], 0));
        CK(adamw_decay_range(e, m, v, e->lo[l_lo].wqkv, l_hi < NL ? e->lo[l_hi].wqkv : e->wp, e->opt_side));
    } else {
        CK((int)hipEventRecord(e->opt_ev[seg], e->opt_side));
        CK((int)hipStreamWaitEvent(st, e->opt_ev[seg], 0));
    }
    return MB_OK;
}

int mb_bert_train_step(mb_bert_engine* e, const int64_t* input_ids, const float* visual, const float* acoustic,
                       const int64_t* attention_mask, const int64_t* token_type_ids, const float* labels, int B, int L,
                       uint64_t seed, uint64_t step, float* logits, float* loss, float* loss_run, float* m, float* v, float lr,
                       float beta1, float beta2, float eps, float weight_decay, int opt_step, int correct_bias, float grad_scale,
                       float loss_scale, int mode, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (!e || !e->P || !e->G || !e->ws) return MB_ERR_ARG;
    const mb_bert_config& c = e->c;
    if (B < 1 || B > c.max_batch || L < 1 || L > c.max_seq) return MB_ERR_SHAPE;
    if (!input_ids || !visual || !acoustic || !attention_mask || !token_type_ids || !labels || !logits || !loss) return MB_ERR_ARG;
    if ((m == nullptr) != (v == nullptr) || (mode != 1 && mode != 2)) return MB_ERR_ARG;
    const int T = B * L;
    char* ws = e->ws;
    CK(ensure_side(e));
    e->training = 1;
    CK(prepare_pass(e, T, st));
    int nseg = 1;
    if (m && e->opt_chunk > 0 && c.num_layers % e->opt_chunk == 0 && !e->overlap_wgrad && !e->prof) {
        nseg = c.num_layers / e->opt_chunk + 1;
        if (!e->opt_side) {
            CK((int)hipStreamCreateWithFlags(&e->opt_side, hipStreamNonBlocking));
            e->opt_ev.assign((size_t)c.num_layers + 1, nullptr);
            for (auto& ev : e->opt_ev) CK((int)hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        }
    }
    return train_step_impl(e, ws, c.visual_dim, c.acoustic_dim, c.num_labels, input_ids, visual, acoustic, attention_mask, token_type_ids,
                           labels, B, L, seed, step, logits, loss, loss_run, m, v, lr, beta1, beta2, eps, weight_decay, opt_step,
                           correct_bias, grad_scale, loss_scale, mode, e->prof, st,
                           [&](int sg, float* lg, float* ls, float* lr_, float* m_, float* v_, float sc, hipStream_t s) {
                               return enqueue_step(e, sg, nseg, B, L, lg, ls, lr_, m_, v_, sc, s);
                           },
                           nseg, [&](int sg, hipStream_t s) { return between_segments(e, sg, nseg, m, v, s); });
}

// ------------------------------------------------------------------------------------------------ data-parallel step, one call
// mb_bert_train_step with the gradient exchange inside (include/magbert_hip.h, csrc/comm.hip).  `plan` = layers per backward segment
// (dp_chunk_plan: 4, 4, 2, 2).  Segments, each a LINEAR graph:
//   s < nb      : (s = 0: forward + head) + the backward of plan[s] layers (+ s = nb-1: MAG + embeddings + the ONE LayerNorm / bias
//                 reduction)                 -> between: all-reduce of those layers' GEMM weights (the last one: + the tail's exchange)
//   nb          : AdamW over the GEMM weights of every layer but the last segment's   (waits for the early pieces only)
//   nb + 1      : AdamW over the last segment's layers, the rest of the decay slab and the no-decay slab   (waits for everything)
// AdamW over [b, en) of the decay slab inside a data-parallel step.  Sharded update (comm->shard): of every chunk of layer GEMM weights
// inside the range only this rank's slice and the replicated remainder are updated (csrc/comm.h: dp_shard_slice); the other slices'
// gradient copies were not reduced here -- they are dead, and cleared unless the next backward overwrites them anyway.
static int adamw_decay_range_dp(mb_bert_engine* e, const mb_comm* comm, const DpSpec& sp, float* m, float* v, size_t b, size_t en, hipStream_t st) {
    if (!comm->shard) return adamw_decay_range(e, m, v, b, en, st);
    std::vector<std::pair<size_t, size_t>> ch(sp.chunk.begin(), sp.chunk.begin() + sp.n_sharded);      // (the rest is replicated)
    std::sort(ch.begin(), ch.end());
    size_t cur = b;
    ZeroRanges dead = {};
    for (const auto& c : ch) {
        if (c.second <= cur || c.first >= en) continue;
        if (c.first < cur || c.second > en) return MB_ERR_MODE;          // (ranges are unions of whole chunks)
        CK(adamw_decay_range(e, m, v, cur, c.first, st));
        const ShardSlice sl = dp_shard_slice(comm, c.first, c.second);
        CK(adamw_decay_range(e, m, v, sl.mine_b, sl.mine_e, st));
        CK(adamw_decay_range(e, m, v, sl.rem_b, sl.rem_e, st));
        if (!e->keep_in_step()) {
            if (dead.n + 2 > MB_ZERO_MAX) { CK(zero_fill_ranges(dead, st)); dead = ZeroRanges{}; }       // (add() drops what does not fit)
            dead.add(e->G + c.first, (sl.mine_b - c.first) * 4);
            dead.add(e->G + sl.mine_e, (sl.rem_b - sl.mine_e) * 4);
        }
        cur = c.second;
    }
    if (dead.n) CK(zero_fill_ranges(dead, st));
    return adamw_decay_range(e, m, v, cur, en, st);
}

static int enqueue_step_dp(mb_bert_engine* e, int seg, const std::vector<int>& plan, const mb_comm* comm, const DpSpec& sp, int B, int L,
                           float* logits, float* loss, float* loss_run, float* m, float* v, float loss_scale, hipStream_t st) {
    char* ws = e->ws;
    const int NL = e->c.num_layers, nb = (int)plan.size();
    const float* lab = (const float*)(ws + e->ws_in_lab);
    float* keep_attn = e->attn_out;
    e->attn_out = nullptr;
    struct Restore { mb_bert_engine* e; float* p; ~Restore() { e->attn_out = p; } } restore{e, keep_attn};
    if (e->head_mask || e->emb_in || e->pos_ids) return MB_ERR_MODE;
    // sharded update with several pieces: nf forward-only segments in front (piece k = the layers of chunk nb-1-k, lowest first); the
    // last piece -- the top layers + head -- stays in front of the first backward segment
    const int nf = comm->nf;
    auto layers_of = [&](int chunk, int& l0, int& l1) { l1 = NL; for (int s = 0; s < chunk; ++s) l1 -= plan[s]; l0 = l1 - plan[chunk]; };
    if (seg <= nf) {
        int l0 = 0, l1 = NL;
        if (nf > 0) layers_of(nb - 1 - seg, l0, l1);
        CK(bert_forward_range(e, (const int64_t*)(ws + e->ws_in_ids), (const float*)(ws + e->ws_in_vis), (const float*)(ws + e->ws_in_aco),
                              (const int64_t*)(ws + e->ws_in_mask), (const int64_t*)(ws + e->ws_in_seg), lab, B, L, 1, 0, 0, logits, loss,
                              loss_run, st, l0, l1, seg == 0, seg == nf));
        if (seg < nf) return MB_OK;
    }
    seg -= nf;
    if (seg < nb) {
        int done = 0;
        for (int s = 0; s < seg; ++s) done += plan[s];
        // backward stages: 0 = head, 1 .. NL = layers NL-1 .. 0, NL+1 = MAG + embeddings
        return mb_bert_backward(e, nullptr, lab, loss_scale, seg == 0 ? 0 : 1 + done, seg == nb - 1 ? NL + 2 : 1 + done + plan[seg], st);
    }
    const AdamArgs none = {};
    const size_t nd = e->n_decay, n = e->n_params;
    const size_t split = e->lo[plan[nb - 1] < NL ? plan[nb - 1] : 0].wqkv;      // first GEMM weight of the layers reduced early
    if (seg == nb) {
        CK(e->prof_mark(2 * NL, st));
        if (nb > 1) return adamw_decay_range_dp(e, comm, sp, m, v, split, e->wp, st);
        // (one backward segment: no early range -- dp_between waited for everything -- so this segment takes the no-decay slab)
        return adamw_step(e->P + nd, e->G + nd, m + nd, v + nd, nullptr, n - nd, 0, 0, 0, none, 1, st, e->adam_state(ws) + 1);
    }
    CK(adamw_decay_range_dp(e, comm, sp, m, v, 0, nb > 1 ? split : e->wp, st));
    CK(adamw_decay_range(e, m, v, e->wp, nd, st));
    if (nb > 1) CK(adamw_step(e->P + nd, e->G + nd, m + nd, v + nd, nullptr, n - nd, 0, 0, 0, none, 1, st, e->adam_state(ws) + 1));
    return e->prof_mark(2 * NL + 1, st);
}

int mb_bert_train_step_dp(mb_bert_engine* e, const int64_t* input_ids, const float* visual, const float* acoustic,
                          const int64_t* attention_mask, const int64_t* token_type_ids, const float* labels, int B, int L,
                          uint64_t seed, uint64_t step, float* logits, float* loss, float* loss_run, float* m, float* v, float lr,
                          float beta1, float beta2, float eps, float weight_decay, int opt_step, int correct_bias, float grad_scale,
                          float loss_scale, int mode, void* stream, mb_comm* comm) {
    hipStream_t st = (hipStream_t)stream;
    if (!e || !e->P || !e->G || !e->ws || !comm) return MB_ERR_ARG;
    const mb_bert_config& c = e->c;
    if (B < 1 || B > c.max_batch || L < 1 || L > c.max_seq) return MB_ERR_SHAPE;
    if (!input_ids || !visual || !acoustic || !attention_mask || !token_type_ids || !labels || !logits || !loss) return MB_ERR_ARG;
    if (!m || !v || (mode != 1 && mode != 2)) return MB_ERR_ARG;
    if (e->overlap_wgrad || !e->grouped) return MB_ERR_MODE;      // the exchange's pieces assume a layer's weight gradients are final when its stage returns
    const int NL = c.num_layers;
    const std::vector<int> plan = dp_chunk_plan(NL);
    const int nb = (int)plan.size();
    DpSpec sp;
    for (int s = 0, hi = NL; s < nb; ++s) {            // segment s finishes layers [hi - plan[s], hi)
        const int lo_l = hi - plan[s];
        sp.chunk.push_back({e->lo[lo_l].wqkv, hi < NL ? e->lo[hi].wqkv : e->wp});
        hi = lo_l;
    }
    sp.tail_begin = e->wp; sp.tail_end = e->n_params;
    sp.word_off = e->word; sp.word_rows = c.vocab_size; sp.H = c.hidden_size;
    sp.ids = (const int64_t*)(e->ws + e->ws_in_ids); sp.T = B * L;
    sp.n_sharded = dp_sharded_chunks(comm, nb);
    const int nf = dp_forward_segments(comm, nb);
    comm->nf = nf;
    if (comm->shard) {
        // what the next forward reads of a layer's GEMM weights: their bf16 shadow (bf16 mode) or the fp32 parameters themselves
        const bool bf = c.dtype == DT_BF16;
        for (const auto& ch : sp.chunk)
            if (bf && (!e->SH || ch.first < e->sh_begin || ch.second > e->sh_end)) return MB_ERR_MODE;
        sp.gather_base = bf ? (char*)e->SH : (char*)e->P; sp.gather_es = bf ? 2 : 4;
    }
    CK(dp_step_begin(comm, st, nf > 0));          // (sharded update: the previous step's all-gathers -- cut mode: awaited piece by piece)
    e->training = 1;
    CK(prepare_pass(e, B * L, st));
    // (the plan is part of the graphs' identity: nseg alone would not tell 4,4,2,2 from 2,2,4,4)
    int variant = 1;
    for (int x : plan) variant = variant * 13 + x;
    variant = (variant * 4 + comm->event_mode) * 2 + (comm->shard ? 1 : 0);
    return train_step_impl(e, e->ws, c.visual_dim, c.acoustic_dim, c.num_labels, input_ids, visual, acoustic, attention_mask, token_type_ids,
                           labels, B, L, seed, step, logits, loss, loss_run, m, v, lr, beta1, beta2, eps, weight_decay, opt_step,
                           correct_bias, grad_scale, loss_scale, mode, e->prof, st,
                           [&](int sg, float* lg, float* ls, float* lr_, float* m_, float* v_, float sc, hipStream_t s) {
                               CK(dp_segment_begin(comm, nb, sg, s));
                               CK(enqueue_step_dp(e, sg, plan, comm, sp, B, L, lg, ls, lr_, m_, v_, sc, s));
                               CK(dp_segment_end(comm, nb, sg, s));
                               return (int)MB_OK;
                           },
                           nf + nb + 2, [&](int sg, hipStream_t s) { return dp_between(comm, sp, e->G, sg, s); }, variant, comm, dp_finish_segment_graph);
}

// ------------------------------------------------------------------------------------------------ stage-driven step (data parallel)
// The same step cut at its backward stages, for a host that issues the gradient exchange between them (one RCCL all-reduce piece
// per stage on a side stream): mb_bert_stage_forward = step prologue (batch gather, dropout keys) + the forward + MSE as ONE
// replayed graph, mb_bert_stage_backward(stage) = that stage as one replayed graph.  16 graph launches per step instead of ~210
// kernel launches from a host that is the bottleneck of the launch-by-launch form (4.0 ms of host time per 4.4 ms step inside a
// torch process, profiles/r03_dp_force.txt).  The optimizer stays with the caller (mb_adamw_step on the reduced gradients).
int mb_bert_stage_forward(mb_bert_engine* e, const int64_t* input_ids, const float* visual, const float* acoustic,
                          const int64_t* attention_mask, const int64_t* token_type_ids, const float* labels, int B, int L,
                          uint64_t seed, uint64_t step, float* logits, float* loss, float* loss_run, int mode, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (!e || !e->P || !e->G || !e->ws) return MB_ERR_ARG;
    const mb_bert_config& c = e->c;
    if (B < 1 || B > c.max_batch || L < 1 || L > c.max_seq) return MB_ERR_SHAPE;
    if (!input_ids || !visual || !acoustic || !attention_mask || !token_type_ids || !labels || !logits || !loss) return MB_ERR_ARG;
    if (mode != 1 && mode != 2) return MB_ERR_ARG;
    if (e->head_mask || e->emb_in || e->pos_ids || e->deferred) return MB_ERR_MODE;
    if (e->prof) mode = 2;              // timing events around kernels: launch by launch (events cannot live inside a captured graph)
    char* ws = e->ws;
    e->training = 1;
    CK(prepare_pass(e, B * L, st));
    PrologueArgs pa = {};
    e->fill_copies(pa, ws, input_ids, visual, acoustic, attention_mask, token_type_ids, labels, B, L, c.visual_dim, c.acoustic_dim, c.num_labels);
    pa.seed = seed; pa.step = step; pa.keys = e->key_state(ws); pa.nsites = e->nsites;
    CK(step_prologue(pa, st));
    float* keep_attn = e->attn_out;
    e->attn_out = nullptr;
    struct Restore { mb_bert_engine* e; float* p; ~Restore() { e->attn_out = p; } } restore{e, keep_attn};
    return e->run_stage_graph(B, L, -1, 0, logits, loss, loss_run, 0.f, mode, st, [&](hipStream_t s) {
        return mb_bert_forward(e, (const int64_t*)(ws + e->ws_in_ids), (const float*)(ws + e->ws_in_vis), (const float*)(ws + e->ws_in_aco),
                               (const int64_t*)(ws + e->ws_in_mask), (const int64_t*)(ws + e->ws_in_seg), (const float*)(ws + e->ws_in_lab), B, L, 1,
                               0, 0, logits, loss, loss_run, s);
    });
}
int mb_bert_stage_backward(mb_bert_engine* e, float loss_scale, int stage, int mode, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (!e || !e->G || !e->ran_forward || !e->training) return MB_ERR_ARG;
    if (stage < 0 || stage > e->c.num_layers + 1 || (mode != 1 && mode != 2)) return MB_ERR_ARG;
    if (e->deferred) return MB_ERR_MODE;
    if (e->prof) mode = 2;
    if (stage == 0) CK(e->begin_backward_pass(e->G, st));        // outside the graph: decides store vs accumulate (part of the graph's identity)
    const float* lab = (const float*)(e->ws + e->ws_in_lab);
    return e->run_stage_graph(e->B, e->L, stage, e->ow_pass ? 1 : 0, e->logits, nullptr, nullptr, loss_scale, mode, st, [&](hipStream_t s) {
        return mb_bert_backward(e, nullptr, lab, loss_scale, stage, stage + 1, s);
    });
}

const int64_t* mb_bert_staged_input_ids(const mb_bert_engine* e) { return (e && e->ws) ? (const int64_t*)(e->ws + e->ws_in_ids) : nullptr; }

int mb_bert_load_batch(mb_bert_engine* e, const int64_t* input_ids, const float* visual, const float* acoustic,
                       const int64_t* attention_mask, const int64_t* token_type_ids, const float* labels, int B, int L,
                       const void** staged6, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (!e || !e->ws || !staged6) return MB_ERR_ARG;
    const mb_bert_config& c = e->c;
    if (B < 1 || B > c.max_batch || L < 1 || L > c.max_seq) return MB_ERR_SHAPE;
    if (!input_ids || !visual || !acoustic || !attention_mask || !token_type_ids) return MB_ERR_ARG;
    const int T = B * L;
    char* ws = e->ws;
    CK(prepare_pass(e, T, st));          // the one-time clearing of the workspace must not land on top of the staged batch
    PrologueArgs pa = {};
    e->fill_copies(pa, ws, input_ids, visual, acoustic, attention_mask, token_type_ids, labels, B, L, c.visual_dim, c.acoustic_dim,
                   c.num_labels);
    CK(step_prologue(pa, st));
    e->staged(ws, labels != nullptr, staged6);
    return MB_OK;
}

int mb_bert_graph_stats(const mb_bert_engine* e, size_t* captures, size_t* launches) {
    if (!e) return MB_ERR_ARG;
    if (captures) *captures = e->graph_captures;
    if (launches) *launches = e->graph_launches;
    return MB_OK;
}

int mb_bert_set_profiling(mb_bert_engine* e, int on) {
    if (!e) return MB_ERR_ARG;
    return e->set_profiling(on, e->c.num_layers);
}
int mb_bert_profile_wgrad_us(mb_bert_engine* e, float* avg_us) {
    if (!e || !e->grouped) return MB_ERR_ARG;
    return e->prof_span_us(0, e->c.num_layers, avg_us);
}
int mb_bert_profile_adamw_us(mb_bert_engine* e, float* us) {
    if (!e) return MB_ERR_ARG;
    return e->prof_span_us(2 * e->c.num_layers, 1, us);
}

// ---- optional outputs and the autograd edge of the base model (bert.py:147-156, 227-237)
const void* mb_bert_hidden_state(const mb_bert_engine* e, int i) {
    if (!e || !e->ws || i < 0 || i > e->c.num_layers) return nullptr;
    return e->ws + e->ws_x[i];
}
int mb_bert_set_attention_output(mb_bert_engine* e, float* probs) {
    if (!e) return MB_ERR_ARG;
    e->attn_out = probs;
    return MB_OK;
}
int mb_bert_mark_grads_zero(mb_bert_engine* e, int known_zero) {
    if (!e) return MB_ERR_ARG;
    e->grads_zero = known_zero != 0;
    if (known_zero) e->grads_stale = false;       // the caller zeroed the whole buffer itself
    return MB_OK;
}
int mb_bert_materialize_grads(mb_bert_engine* e, void* stream) {
    if (!e) return MB_ERR_ARG;
    return e->materialize_grads(e->G, (hipStream_t)stream);
}
int mb_bert_grads_stale(const mb_bert_engine* e) { return e && e->grads_stale ? 1 : 0; }
int mb_bert_set_head_mask(mb_bert_engine* e, const float* head_mask) {
    if (!e) return MB_ERR_ARG;
    e->head_mask = head_mask;
    return MB_OK;
}
int mb_bert_set_inputs_embeds(mb_bert_engine* e, const float* inputs_embeds) {
    if (!e) return MB_ERR_ARG;
    e->emb_in = inputs_embeds;
    return MB_OK;
}
int mb_bert_set_position_ids(mb_bert_engine* e, const int64_t* position_ids) {
    if (!e) return MB_ERR_ARG;
    e->pos_ids = position_ids;
    return MB_OK;
}
const float* mb_bert_inputs_embeds_grad(const mb_bert_engine* e) {
    if (!e || !e->ws || !e->ran_forward) return nullptr;
    return (const float*)(e->ws + e->ws_dsum);
}
int mb_bert_backward_outputs(mb_bert_engine* e, const void* d_sequence_output, const void* d_pooler_preact, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (!e || !e->G || !e->ran_forward) return MB_ERR_ARG;
    CK(e->begin_backward_pass(e->G, st));
    const mb_bert_config& c = e->c;
    const int dt = c.dtype, H = c.hidden_size, B = e->B, L = e->L, T = B * L, NL = c.num_layers;
    char* ws = e->ws;
    const size_t bytes = (size_t)T * H * esize(dt);
    if (d_sequence_output) CK((int)hipMemcpyAsync(ws + e->ws_dxa, d_sequence_output, bytes, hipMemcpyDeviceToDevice, st));
    else CK(zero_fill(ws + e->ws_dxa, bytes, st));
    if (d_pooler_preact) {
        // pooler Linear (bert.py:230-231 -> BertPooler): weight / bias gradients, then its input gradient lands on the [CLS] rows
        const char* xf = ws + e->ws_x[NL];
        CK(gemm(dt, GEMM_TN, EPI_ACCUM_F32, H, H, B, d_pooler_preact, H, xf, L * H, nullptr, H, nullptr, e->G + e->wp, nullptr, nullptr, 0,
                kNoDrop, 1, 64, st));
        CK(colsum(dt, d_pooler_preact, H, e->G + e->bp, B, H, st, e->acc_of(ws, e->G)));
        CK(gemm(dt, GEMM_NN, EPI_ADD_RES, B, H, H, d_pooler_preact, H, e->W(e->wp), H, ws + e->ws_dxa, L * H, nullptr, nullptr, nullptr,
                ws + e->ws_dxa, L * H, kNoDrop, 1, 64, st));
    }
    return MB_OK;
}

const void* mb_bert_sequence_output(const mb_bert_engine* e) { return e->ws ? e->ws + e->ws_x[e->c.num_layers] : nullptr; }
const float* mb_bert_pooled_output(const mb_bert_engine* e) { return e->ws ? (const float*)(e->ws + e->ws_head_pooled) : nullptr; }

int mb_bert_stage_grad_ranges(const mb_bert_engine* e, int stage, size_t* offs, size_t* lens, int cap) {
    const int NL = e->c.num_layers;
    std::vector<std::pair<size_t, size_t>> r;
    auto span = [&](size_t a, size_t b) { r.push_back({a, b - a}); };
    if (stage == 0) {
        span(e->wp, e->sh_end);                               // pooler weight
        span(e->wc, e->n_decay);                              // classifier weight
        span(e->bp, e->mag_bhv);                              // pooler bias
        span(e->bc, e->n_params);                             // classifier bias
    } else if (stage <= NL) {
        const int l = NL - stage;
        const LayerOff& o = e->lo[l];
        auto wspan = [&](int k) { span(e->lo[k].wqkv, k + 1 < NL ? e->lo[k + 1].wqkv : e->wp); };
        if (!e->deferred) wspan(l);
        else if (l + 1 < NL) wspan(l + 1);       // deferred join: the weights arrive one stage late
        span(o.bqkv, l + 1 < NL ? e->lo[l + 1].bqkv : e->emb_lnw);
    } else if (stage == NL + 1) {
        if (e->deferred) span(e->lo[0].wqkv, NL > 1 ? e->lo[1].wqkv : e->wp);
        span(e->word, e->wc);                                 // embeddings + MAG weights
        span(e->emb_lnw, e->bp);                              // embeddings LayerNorm
        span(e->mag_bhv, e->bc);                              // MAG biases + LayerNorm
    } else return -1;
    int n = 0;
    for (auto& p : r) { if (n < cap) { offs[n] = p.first; lens[n] = p.second; } ++n; }
    return n;
}

}  // extern "C"
