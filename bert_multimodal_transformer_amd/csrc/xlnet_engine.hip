// Native step executor for MAG_XLNetForSequenceClassification (/root/reference/xlnet.py:432-527 -> :15-429 ->
// /root/reference/modeling.py:25-51), BASELINE.json config 4.  Same contract as engine.hip: one C call enqueues a whole pass,
// caller-owned flat parameter / gradient / bf16-shadow / workspace buffers, reference state-dict names.
//
// Only the configuration the reference driver exercises is built (xlnet-base-cased: attn_type "bi", no mems / perm_mask /
// target_mapping; multimodal_driver.py:363-370), sequence length <= 128 (MOSI: 50).
//
// Flat layout:  [ decay | no-decay | frozen ]
//   decay    : per layer rel_attn.{q,k,v,o,r} ([d_model][n_head*d_head], consumed k-major as stored), ff.layer_1.weight,
//              ff.layer_2.weight ; sequence_summary.summary.weight            <- bf16 shadow range
//              per layer rel_attn.seg_embed, rel_attn.layer_norm.weight, ff.layer_norm.weight (XLNet's LayerNorm is called
//              `layer_norm`, so its weight IS decayed by the driver's substring rule, multimodal_driver.py:329-343) ;
//              word_embedding ; MAG weights ; logits_proj.weight
//   no-decay : per layer r_r_bias, r_s_bias, r_w_bias, both layer_norm.bias, ff biases ; MAG biases + MAG.LayerNorm.* ;
//              summary.bias ; logits_proj.bias
//   frozen   : transformer.mask_emb (never receives a gradient in this configuration; HF AdamW skips grad-less parameters)
#include "engine_common.h"
#include "comm.h"

struct XlLayerOff { size_t q, k, v, o, r, w1, w2, seg, ralnw, fflnw, rrb, rsb, rwb, ralnb, fflnb, b1, b2; };
struct XlLayerWs { size_t qkv, kr, vec, psave, s1, st1, y1, u, g, s2, st2; };

struct mb_xlnet_engine : StepMixin {
    mb_xlnet_config c;
    std::vector<TensorInfo> tensors;
    std::vector<XlLayerOff> lo;
    size_t word, wsum, bsum, wc, bc, mask_emb, small_decay_begin;
    size_t mag_whv, mag_wha, mag_wv, mag_wa, mag_bhv, mag_bha, mag_bv, mag_ba, mag_lnw, mag_lnb;
    size_t n_params, n_trainable, n_decay, sh_begin, sh_end;
    MagWs mw;
    size_t ws_mag, ws_magout, ws_pos, ws_xs, ws_head_z, ws_head_pooled;
    std::vector<size_t> ws_x;
    std::vector<XlLayerWs> lw;
    size_t ws_dsa[2], ws_dzda[2], ws_dsb[2], ws_dzdb[2], ws_du[2], ws_dqkv[2], ws_dkr[2];   // dY operands of the weight gradients: ping-pong by layer parity
    size_t ws_dxa, ws_dxb, ws_dvec, ws_gsave, ws_dz, ws_dxs, ws_lnp_a, ws_lnp_b;
    size_t ws_rhalf = 0;           // fp32 [H][H]: the second k-half of a layer's relative-position weight gradient (see mb_xlnet_backward)
    int split_r = 1;               // MB_XL_SPLIT_R=0: the r problem of the grouped launch keeps its whole K = 2 x tokens (round-3 form)
    size_t lnp_stride = 0;         // floats per layer in each of the two LayerNorm partial buffers
    int mag_nblk = 0;              // slabs MAG's gate backward wrote into slot n_layer
    int prefetch = 1;              // MB_PREFETCH=0: the LayerNorm kernels do not touch the next GEMMs' weights (common.h Prefetch)
    const float* head_mask = nullptr;   // mb_xlnet_set_head_mask: [n_layer][n_head] fp32 (caller-owned device memory)
    const char* mems = nullptr; int mlen = 0;      // mb_xlnet_set_mems: [n_layer][B][mlen][d_model], activation dtype (caller-owned)
    const uint8_t* perm = nullptr;      // mb_xlnet_set_perm_mask: [B][L][L] bytes, != 0 <=> query i may not attend to key j (xlnet.py:265-296)
    const float* emb_in = nullptr;      // mb_xlnet_set_inputs_embeds: [B*L][H] fp32 word embeddings given instead of input_ids (xlnet.py:306-313)
    size_t ws_demb = 0;                 // fp32 [T][H]: gradient of the given embeddings
    bool ran_forward = false;
    bool fuse_qkv = true;               // MB_XL_FUSE_QKV=0: q, k, v as three GEMMs each way (round-2 form, kept for A/B runs)
    size_t ws_bytes;
    float* P = nullptr; float* G = nullptr; char* SH = nullptr; char* ws = nullptr;
    const int64_t* ids = nullptr; const int64_t* seg = nullptr; const int64_t* mask = nullptr;
    int B = 0, L = 0, training = 0, padT = -1;
    int group_wgrad = 256;         // MB_GROUP_WGRAD: tile of the per-layer grouped weight-gradient launch (64 | 128 | 256 = 256 x 128 ping-pong), 0 = one by one
                                   // (256 since the ping-pong loop has one barrier per stage: 3.895 -> 3.872 ms, profiles/r06_ride_budget3.txt; fp32 / odd widths: 128)
    // MB_OVERLAP_WGRAD=1: the grouped launch of layer l runs on an internal side stream under the dgrad chain of layer l-1 (round-1
    // default; measured equal to the in-line launch on the MAG-BERT engine, which keeps the step one in-order, graph-friendly sequence)
    int overlap_wgrad = 0;
    bool deferred = false;
    hipStream_t side = nullptr;
    std::vector<hipEvent_t> evs;   // [2 * n_layer]: fork, done
    // AdamW riders (kernels.h AdamRide; as engine.hip).  This engine's grouped weight gradient fills the chip (504 tiles in 512 slots), so
    // the hosts are the ffn1 / ffn2 / out dgrad launches and the relative-attention backward: MB_ADAMW_RIDE=0 turns them off
    int adam_ride = 1;
    RideOpts ride_opts;
    int ride_attn = 1, ride_attn_blocks = 0;
    long ride_attn_params = 0;
    float* ride_m = nullptr; float* ride_v = nullptr;
    size_t ride_cursor = 0;
    bool ws_zeroed = false;
    uint64_t seed = 0, step = 0;
    float* logits = nullptr;

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
    const void* W(size_t off) const { return c.dtype == DT_BF16 ? (const void*)(SH + off * 2) : (const void*)(P + off); }
    DropKey key(uint32_t site, float p) const { return step_key(ws, training != 0, seed, step, site, p); }
};

// dropout sites: 0 word embedding, 1 MAG, 2 summary-last, 3 final output, 4 pos_emb ; layer l: 16+8l+{0 attn probs, 1 attn out,
// 2 ff activation, 3 ff out}
enum { XS_EMB = 0, XS_MAG = 1, XS_HEAD = 2, XS_FINAL = 3, XS_POS = 4, XS_LAYER0 = 16 };

static void xl_build_layout(mb_xlnet_engine* e) {
    const mb_xlnet_config& c = e->c;
    const int64_t H = c.d_model, I = c.d_inner, V = c.visual_dim, A = c.acoustic_dim, nh = c.n_head, dh = H / nh;
    size_t cur = 0;
    e->lo.resize(c.n_layer);
    char buf[128];
    auto nm = [&](int l, const char* s) { snprintf(buf, sizeof buf, "transformer.layer.%d.%s", l, s); return std::string(buf); };
    e->sh_begin = 0;
    for (int l = 0; l < c.n_layer; ++l) {
        XlLayerOff& o = e->lo[l];
        o.q = e->add(nm(l, "rel_attn.q"), {H, nh, dh}, 1, cur);
        o.k = e->add(nm(l, "rel_attn.k"), {H, nh, dh}, 1, cur);
        o.v = e->add(nm(l, "rel_attn.v"), {H, nh, dh}, 1, cur);
        o.o = e->add(nm(l, "rel_attn.o"), {H, nh, dh}, 1, cur);
        o.r = e->add(nm(l, "rel_attn.r"), {H, nh, dh}, 1, cur);
        o.w1 = e->add(nm(l, "ff.layer_1.weight"), {I, H}, 1, cur);
        o.w2 = e->add(nm(l, "ff.layer_2.weight"), {H, I}, 1, cur);
    }
    e->wsum = e->add("sequence_summary.summary.weight", {H, H}, 1, cur);
    e->sh_end = cur;
    e->small_decay_begin = cur;
    for (int l = 0; l < c.n_layer; ++l) {
        XlLayerOff& o = e->lo[l];
        o.seg = e->add(nm(l, "rel_attn.seg_embed"), {2, nh, dh}, 1, cur);
        o.ralnw = e->add(nm(l, "rel_attn.layer_norm.weight"), {H}, 1, cur);
        o.fflnw = e->add(nm(l, "ff.layer_norm.weight"), {H}, 1, cur);
    }
    e->word = e->add("transformer.word_embedding.weight", {c.vocab_size, H}, 1, cur);
    e->mag_whv = e->add("transformer.MAG.W_hv.weight", {H, V + H}, 1, cur);
    e->mag_wha = e->add("transformer.MAG.W_ha.weight", {H, A + H}, 1, cur);
    e->mag_wv = e->add("transformer.MAG.W_v.weight", {H, V}, 1, cur);
    e->mag_wa = e->add("transformer.MAG.W_a.weight", {H, A}, 1, cur);
    e->wc = e->add("logits_proj.weight", {c.num_labels, H}, 1, cur);
    e->n_decay = cur;
    for (int l = 0; l < c.n_layer; ++l) {
        XlLayerOff& o = e->lo[l];
        o.rrb = e->add(nm(l, "rel_attn.r_r_bias"), {nh, dh}, 0, cur);
        o.rsb = e->add(nm(l, "rel_attn.r_s_bias"), {nh, dh}, 0, cur);
        o.rwb = e->add(nm(l, "rel_attn.r_w_bias"), {nh, dh}, 0, cur);
        o.ralnb = e->add(nm(l, "rel_attn.layer_norm.bias"), {H}, 0, cur);
        o.fflnb = e->add(nm(l, "ff.layer_norm.bias"), {H}, 0, cur);
        o.b1 = e->add(nm(l, "ff.layer_1.bias"), {I}, 0, cur);
        o.b2 = e->add(nm(l, "ff.layer_2.bias"), {H}, 0, cur);
    }
    e->mag_bhv = e->add("transformer.MAG.W_hv.bias", {H}, 0, cur);
    e->mag_bha = e->add("transformer.MAG.W_ha.bias", {H}, 0, cur);
    e->mag_bv = e->add("transformer.MAG.W_v.bias", {H}, 0, cur);
    e->mag_ba = e->add("transformer.MAG.W_a.bias", {H}, 0, cur);
    e->mag_lnw = e->add("transformer.MAG.LayerNorm.weight", {H}, 0, cur);
    e->mag_lnb = e->add("transformer.MAG.LayerNorm.bias", {H}, 0, cur);
    e->bsum = e->add("sequence_summary.summary.bias", {H}, 0, cur);
    e->bc = e->add("logits_proj.bias", {c.num_labels}, 0, cur);
    e->n_trainable = cur;
    e->mask_emb = e->add("transformer.mask_emb", {1, 1, H}, 2, cur);      // decay code 2 = frozen
    e->n_params = cur;

    const size_t es = esize(c.dtype);
    const size_t T = align_up((size_t)c.max_batch * c.max_seq, 64);
    const size_t R = align_up((size_t)c.max_batch * 2 * c.max_seq, 64);
    const size_t LPm = c.max_seq <= 32 ? 32 : (c.max_seq <= 64 ? 64 : 128);
    const size_t PP = (size_t)c.max_batch * nh * LPm * LPm;    // saved probabilities / score gradients: [B * heads][LP][LP], LP = 32 | 64 | 128
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
    e->ws_magout = w.take(T * H * es);
    e->ws_pos = w.take(R * H * es);
    e->ws_x.resize(c.n_layer + 1);
    for (int l = 0; l <= c.n_layer; ++l) e->ws_x[l] = w.take(T * H * es);
    e->lw.resize(c.n_layer);
    for (int l = 0; l < c.n_layer; ++l) {
        XlLayerWs& x = e->lw[l];
        x.qkv = w.take(T * 3 * H * es); x.kr = w.take(R * H * es); x.vec = w.take(T * H * es); x.psave = w.take(PP * es);
        x.s1 = w.take(T * H * es); x.st1 = w.take(2 * T * 4); x.y1 = w.take(T * H * es); x.u = w.take(T * I * es);
        x.g = w.take(T * I * es); x.s2 = w.take(T * H * es); x.st2 = w.take(2 * T * 4);
    }
    e->ws_xs = w.take((size_t)c.max_batch * H * es);
    e->ws_head_z = w.take((size_t)c.max_batch * H * 4);
    e->ws_head_pooled = w.take((size_t)c.max_batch * H * 4);
    e->ws_dxa = w.take(T * H * es); e->ws_dxb = w.take(T * H * es);
    for (int k = 0; k < 2; ++k) {
        e->ws_dsa[k] = w.take(T * H * es); e->ws_dzda[k] = w.take(T * H * es); e->ws_dsb[k] = w.take(T * H * es);
        e->ws_dzdb[k] = w.take(T * H * es); e->ws_du[k] = w.take(T * I * es); e->ws_dqkv[k] = w.take(T * 3 * H * es);
        e->ws_dkr[k] = w.take(R * H * es);
    }
    e->ws_dvec = w.take(T * H * es); e->ws_gsave = w.take(PP * es);
    e->ws_demb = w.take(T * H * 4);
    e->ws_dz = w.take((size_t)c.max_batch * H * es); e->ws_dxs = w.take((size_t)c.max_batch * H * es);
    e->ws_rhalf = w.take((size_t)H * H * 4);
    e->lnp_stride = ln_partials_floats((int)T, (int)H);        // per-layer slabs: the single-call step reduces all layers at once
    e->ws_lnp_a = w.take(e->lnp_stride * 4 * (c.n_layer + 1)); e->ws_lnp_b = w.take(e->lnp_stride * 4 * (c.n_layer + 1));      // (+1: MAG's gate)
    e->carve_step(w, T, (int)V, (int)A, c.max_batch, c.num_labels, XS_LAYER0 + 8 * c.n_layer);
    if (e->deterministic) {          // 64-bit shadow of everything behind the layers' GEMM weights (those have ONE writer per element)
        e->det_begin = e->wsum; e->det_end = e->n_trainable;
        e->ws_det = w.take((e->det_end - e->det_begin) * sizeof(long long));
    }
    e->ws_bytes = w.off;
}

// state the next pass relies on but that is not part of the pass itself (kept out of captured step graphs)
static int xl_prepare_pass(mb_xlnet_engine* e, int T, hipStream_t st) {
    if (e->deferred && e->side)      // a backward that was not run to its last stage may still have weight-gradient GEMMs in flight
        for (size_t l = 0; l < 2 && 2 * l + 1 < e->evs.size(); ++l) CK((int)hipStreamWaitEvent(st, e->evs[2 * l + 1], 0));
    if (!e->ws_zeroed) { CK((int)hipMemsetAsync(e->ws, 0, e->ws_bytes, st)); e->ws_zeroed = true; e->padT = T; }
    if (e->padT != T) {
        // another batch shape ran before: the pad rows [T, Tp) / [2T, Rp) of every buffer a weight gradient reads as its k-major
        // operand may hold stale tokens (kernels never write rows past the batch) -> clear those rows, not the whole workspace
        const mb_xlnet_config& c = e->c;
        const size_t es = esize(c.dtype), H = c.d_model, I = c.d_inner;
        const size_t Tp = align_up((size_t)T, 64), R = (size_t)2 * T, Rp = align_up(R, 64);
        char* ws = e->ws;
        auto zp = [&](size_t off, size_t cols, size_t rows, size_t rows_p) {
            return rows_p > rows ? (int)hipMemsetAsync(ws + off + rows * cols * es, 0, (rows_p - rows) * cols * es, st) : 0;
        };
        CK(zp(e->ws_magout, H, T, Tp)); CK(zp(e->ws_pos, H, R, Rp));
        CK(mag_clear_pad_rows(c.dtype, ws + e->ws_mag, e->mw, T, (int)H, st));
        for (int l = 0; l <= c.n_layer; ++l) CK(zp(e->ws_x[l], H, T, Tp));
        for (int l = 0; l < c.n_layer; ++l) { CK(zp(e->lw[l].vec, H, T, Tp)); CK(zp(e->lw[l].y1, H, T, Tp)); CK(zp(e->lw[l].g, I, T, Tp)); }
        for (int k = 0; k < 2; ++k) {
            CK(zp(e->ws_dsa[k], H, T, Tp)); CK(zp(e->ws_dzda[k], H, T, Tp)); CK(zp(e->ws_dsb[k], H, T, Tp)); CK(zp(e->ws_dzdb[k], H, T, Tp));
            CK(zp(e->ws_du[k], I, T, Tp)); CK(zp(e->ws_dqkv[k], 3 * H, T, Tp)); CK(zp(e->ws_dkr[k], H, R, Rp));
        }
    }
    e->padT = T;
    return MB_OK;
}

extern "C" {

int mb_xlnet_create(const mb_xlnet_config* cfg, mb_xlnet_engine** out) {
    if (!cfg || !out) return MB_ERR_ARG;
    if (cfg->d_model != 768 || cfg->n_head * 64 != cfg->d_model) return MB_ERR_SHAPE;
    if (cfg->d_inner % 128 || cfg->max_seq < 1 || cfg->max_seq > 128 || cfg->max_batch < 1 || cfg->num_labels < 1) return MB_ERR_SHAPE;
    if (cfg->injection_index < 0 || cfg->injection_index >= cfg->n_layer) return MB_ERR_ARG;
    if (cfg->dtype != DT_F32 && cfg->dtype != DT_BF16) return MB_ERR_DTYPE;
    mb_xlnet_engine* e = new mb_xlnet_engine();
    if (const char* v = getenv("MB_GROUP_WGRAD")) e->group_wgrad = atoi(v);
    if (e->group_wgrad == 256 && (cfg->dtype != DT_BF16 || cfg->d_model % 256 != 0 || cfg->d_inner % 256 != 0)) e->group_wgrad = 128;     // 256 x 128 ping-pong tile: bf16, whole tiles
    if (const char* v = getenv("MB_OVERLAP_WGRAD")) e->overlap_wgrad = atoi(v);
    if (const char* v = getenv("MB_WGRAD_OVERWRITE")) e->ow_enable = atoi(v);
    e->deferred = e->overlap_wgrad && (e->group_wgrad == 64 || e->group_wgrad == 128 || e->group_wgrad == 256) && cfg->d_inner % e->group_wgrad == 0 &&
                  cfg->d_model % e->group_wgrad == 0;
    e->c = *cfg;
    if (const char* v = getenv("MB_DETERMINISTIC")) e->deterministic = atoi(v);
    xl_build_layout(e);
    if (const char* v = getenv("MB_XL_FUSE_QKV")) e->fuse_qkv = atoi(v) != 0;
    if (const char* v = getenv("MB_ADAMW_RIDE")) e->adam_ride = atoi(v);
    if (const char* v = getenv("MB_ADAMW_RIDE_DGRAD")) e->ride_opts.dgrad = atoi(v);
    if (const char* v = getenv("MB_ADAMW_RIDE_DGRAD_BLOCKS")) e->ride_opts.dgrad_blocks = atoi(v);
    if (const char* v = getenv("MB_ADAMW_RIDE_DGRAD_PARAMS")) e->ride_opts.dgrad_params = atol(v);
    if (const char* v = getenv("MB_ADAMW_RIDE_DGELU_PARAMS")) e->ride_opts.dgelu_params = atol(v);
    if (const char* v = getenv("MB_ADAMW_RIDE_ATTN")) e->ride_attn = atoi(v);
    if (const char* v = getenv("MB_ADAMW_RIDE_ATTN_BLOCKS")) e->ride_attn_blocks = atoi(v);
    if (const char* v = getenv("MB_ADAMW_RIDE_ATTN_PARAMS")) e->ride_attn_params = atol(v);
    if (const char* v = getenv("MB_XL_SPLIT_R")) e->split_r = atoi(v) != 0;
    if (const char* pv = getenv("MB_PREFETCH")) e->prefetch = atoi(pv);
    if (e->lo[0].k - e->lo[0].q != (size_t)cfg->d_model * cfg->d_model || e->lo[0].v - e->lo[0].k != e->lo[0].k - e->lo[0].q) e->fuse_qkv = false;
    if (const char* v = getenv("MB_ADAMW_KEEP")) e->keep_enable = atoi(v);
    // lazy zeroing (engine_common.h): the seven GEMM weights of every layer, stored by the grouped launches of a pass that may overwrite
    e->ow_covers = (e->group_wgrad == 64 || e->group_wgrad == 128 || e->group_wgrad == 256) && cfg->d_inner % e->group_wgrad == 0 && cfg->d_model % e->group_wgrad == 0;
    e->stale_begin = e->lo[0].q; e->stale_end = e->wsum;
    *out = e;
    return MB_OK;
}
void mb_xlnet_destroy(mb_xlnet_engine* e) {
    if (!e) return;
    if (e->side) hipStreamDestroy(e->side);
    for (auto& ev : e->evs) if (ev) hipEventDestroy(ev);
    e->destroy_prof();
    e->drop_graphs();
    delete e;
}
int mb_xlnet_num_tensors(const mb_xlnet_engine* e) { return (int)e->tensors.size(); }
int mb_xlnet_tensor_info(const mb_xlnet_engine* e, int i, char* name, int name_cap, size_t* offset, size_t* numel, int* ndim,
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
size_t mb_xlnet_param_count(const mb_xlnet_engine* e) { return e->n_params; }
size_t mb_xlnet_decay_count(const mb_xlnet_engine* e) { return e->n_decay; }
void mb_xlnet_shadow_range(const mb_xlnet_engine* e, size_t* b, size_t* en) { *b = e->sh_begin; *en = e->sh_end; }
size_t mb_xlnet_workspace_bytes(const mb_xlnet_engine* e) { return e->ws_bytes; }
int mb_xlnet_bind(mb_xlnet_engine* e, float* params, float* grads, void* shadow, void* workspace, size_t ws_bytes) {
    if (!params || !workspace || ws_bytes < e->ws_bytes) return MB_ERR_ARG;
    if (e->c.dtype == DT_BF16 && !shadow) return MB_ERR_ARG;
    e->P = params; e->G = grads; e->SH = (char*)shadow; e->ws = (char*)workspace;
    e->grads_zero = false;                 // a newly bound gradient buffer: nothing is known about its contents
    e->ws_zeroed = false; e->padT = -1;
    e->drop_graphs();
    char* mws = e->ws + e->ws_mag;
    e->pkw = {e->P + e->mag_whv, e->P + e->mag_wha, e->P + e->mag_wv, e->P + e->mag_wa, mws + e->mw.We, mws + e->mw.Wv, mws + e->mw.Wa,
              MagDims{0, e->c.d_model, e->c.visual_dim, e->c.acoustic_dim, e->mw.Vp, e->mw.Ap}, e->c.dtype};
    return MB_OK;
}
int mb_xlnet_sync_weights(mb_xlnet_engine* e, void* stream) {
    if (!e->P) return MB_ERR_ARG;
    if (e->c.dtype == DT_BF16)
        CK(convert(DT_BF16, e->P + e->sh_begin, e->SH + e->sh_begin * 2, e->sh_end - e->sh_begin, (hipStream_t)stream));
    return MB_OK;
}

// the forward of layers [l0, l1) (`first`: with the pass set-up, the embeddings and the positional table in front; `last`: with the
// summary and the head behind): mb_xlnet_forward is the whole range, the sharded data-parallel step runs it in pieces (engine.hip)
static int xl_forward_range(mb_xlnet_engine* e, const int64_t* input_ids, const float* visual, const float* acoustic,
                            const int64_t* attention_mask, const int64_t* token_type_ids, const float* labels, int B, int L,
                            int training, uint64_t seed, uint64_t step, float* logits, float* loss, float* loss_run, void* stream,
                            int l0, int l1, bool first, bool last) {
    hipStream_t st = (hipStream_t)stream;
    const mb_xlnet_config& c = e->c;
    if (!e->P || !e->ws) return MB_ERR_ARG;
    if (B < 1 || B > c.max_batch || L < 1 || L > c.max_seq) return MB_ERR_SHAPE;
    if ((!input_ids && !e->emb_in) || !visual || !acoustic || !attention_mask || !token_type_ids || !logits) return MB_ERR_ARG;
    if (e->mems && (e->mlen >= L || e->in_step)) return MB_ERR_MODE;      // memories: explicit forwards / backwards only (not the single-call step)
    const int dt = c.dtype, H = c.d_model, I = c.d_inner, T = B * L, nh = c.n_head, R = B * 2 * L;
    const size_t es = esize(dt);
    if (e->emb_in) input_ids = nullptr;           // inputs_embeds given: no table gather, no scatter into the word table
    e->ids = input_ids; e->seg = token_type_ids; e->mask = attention_mask; e->ran_forward = true;
    e->B = B; e->L = L; e->training = training; e->seed = seed; e->step = step; e->logits = logits;
    float* P = e->P;
    char* ws = e->ws;
    if (first && !e->capturing) CK(xl_prepare_pass(e, T, st));
    const float pd = c.dropout;
    if (first) {
        CK(gather_drop_forward(dt, input_ids, e->emb_in ? e->emb_in : P + e->word, ws + e->ws_x[0], T, H, e->key(XS_EMB, pd), st));   // xlnet.py:304-313
        CK(xlnet_pos_emb(dt, ws + e->ws_pos, B, L, H, e->key(XS_POS, pd), st));                                         // xlnet.py:332-333
    }
    for (int l = l0; l < l1; ++l) {
        const XlLayerOff& o = e->lo[l];
        const XlLayerWs& w = e->lw[l];
        const char* xin = ws + e->ws_x[l];
        if (l == c.injection_index) {                                                                                // xlnet.py:371-372
            CK(mag_fwd_impl(dt, xin, visual, acoustic, P + e->mag_whv, P + e->mag_bhv, P + e->mag_wha, P + e->mag_bha,
                            P + e->mag_wv, P + e->mag_bv, P + e->mag_wa, P + e->mag_ba, P + e->mag_lnw, P + e->mag_lnb,
                            c.mag_layer_norm_eps, c.beta_shift, e->key(XS_MAG, c.mag_dropout), ws + e->ws_magout,
                            ws + e->ws_mag, e->mw, T, H, c.visual_dim, c.acoustic_dim, !(e->in_step && e->packed_w), st, true, e->in_step && e->packed));
            xin = ws + e->ws_magout;
        }
        if (e->mems) {
            // cached memories (xlnet.py:81-91, 374-385 -> XLNetRelativeAttention: keys / values over cat([mems[l], h])): the caller laid the
            // segment out as klen = mlen + qlen rows per sample whose first mlen rows are the memory -- their ids / modalities are
            // dummies, their mask is "visible", their segment id 0 (the reference's mem_pad) -- so rows [0, mlen) of every sample of
            // this layer's input are REPLACED by mems[l] here, behind the MAG injection (which the reference applies to h only).  The
            // relative position of query row mlen + i and key row j is (klen - (mlen + i) + j) = qlen - i + j: exactly the index the
            // reference's rel_shift produces for klen keys, so the attention kernels run unchanged on klen rows.  What the layer
            // computes for the memory rows themselves is never read: the next layer replaces them again, the head reads the last row.
            CK((int)hipMemcpy2DAsync((void*)xin, (size_t)L * H * es, e->mems + (size_t)l * B * e->mlen * H * es, (size_t)e->mlen * H * es,
                                     (size_t)e->mlen * H * es, (size_t)B, hipMemcpyDeviceToDevice, st));
        }
        char* qkv = ws + w.qkv;
        // q | k | v | kr projections: x . W with W stored [d_model][n_head*d_head] (einsum "ibh,hnd->ibnd")
        if (e->fuse_qkv) {
            // ONE GEMM, N = 3 H: the three [d_model][n_head * d_head] tensors sit next to each other in the flat buffer (a k-major B
            // whose columns are cut into three segments)
            CK(gemm(dt, GEMM_NN, EPI_ADD_RES, T, 3 * H, H, xin, H, e->W(o.q), H, qkv, 3 * H, nullptr, nullptr, nullptr, nullptr, 0, kNoDrop, 1, 0,
                    st, H, o.k - o.q));
        } else {
            CK(gemm(dt, GEMM_NN, EPI_ADD_RES, T, H, H, xin, H, e->W(o.q), H, qkv, 3 * H, nullptr, nullptr, nullptr, nullptr, 0, kNoDrop, 1, 0, st));
            CK(gemm(dt, GEMM_NN, EPI_ADD_RES, T, H, H, xin, H, e->W(o.k), H, qkv + (size_t)H * es, 3 * H, nullptr, nullptr, nullptr, nullptr, 0, kNoDrop, 1, 0, st));
            CK(gemm(dt, GEMM_NN, EPI_ADD_RES, T, H, H, xin, H, e->W(o.v), H, qkv + (size_t)2 * H * es, 3 * H, nullptr, nullptr, nullptr, nullptr, 0, kNoDrop, 1, 0, st));
        }
        CK(gemm(dt, GEMM_NN, EPI_ADD_RES, R, H, H, ws + e->ws_pos, H, e->W(o.r), H, ws + w.kr, H, nullptr, nullptr, nullptr, nullptr, 0, kNoDrop, 1, 0, st));
        CK(xlnet_attention_forward(dt, qkv, ws + w.kr, P + o.rwb, P + o.rrb, P + o.rsb, P + o.seg, token_type_ids, attention_mask,
                                   ws + w.vec, ws + w.psave, B, L, nh, e->key(XS_LAYER0 + 8 * l + 0, pd), st,
                                   e->head_mask ? e->head_mask + (size_t)l * nh : nullptr, e->perm));
        // post_attention: dropout(vec . o^T) + h -> LayerNorm
        CK(gemm(dt, GEMM_NT, EPI_BIAS_DROP_RES, T, H, H, ws + w.vec, H, e->W(o.o), H, ws + w.s1, H, nullptr, nullptr, nullptr, xin, H,
                e->key(XS_LAYER0 + 8 * l + 1, pd), 1, 0, st));
        // (the LayerNorm launches touch the weights of the GEMMs behind them -- common.h Prefetch: layer_1 | layer_2 here, the next
        //  layer's q | k | v | o | r behind the second one)
        const size_t wes = dt == DT_BF16 ? 2 : 4;
        const Prefetch pf1 = {e->prefetch ? e->W(o.w1) : nullptr, (size_t)2 * I * H * wes, nullptr};
        const Prefetch pf2 = {(e->prefetch && l + 1 < c.n_layer && (l + 1 < l1 || last)) ? e->W(e->lo[l + 1].q) : nullptr, (size_t)5 * H * H * wes, nullptr};      // (not across a forward seam)
        CK(ln_forward(dt, ws + w.s1, P + o.ralnw, P + o.ralnb, c.layer_norm_eps, ws + w.y1, (float*)(ws + w.st1),
                      (float*)(ws + w.st1) + T, T, H, kNoDrop, st, pf1));
        // feed forward: layer_1 -> gelu -> dropout -> layer_2 -> dropout -> LayerNorm(. + inp)
        CK(gemm(dt, GEMM_NT, EPI_BIAS_GELU, T, I, H, ws + w.y1, H, e->W(o.w1), H, ws + w.u, I, ws + w.g, nullptr, P + o.b1, nullptr, 0,
                e->key(XS_LAYER0 + 8 * l + 2, pd), 1, 0, st));
        CK(gemm(dt, GEMM_NT, EPI_BIAS_DROP_RES, T, H, I, ws + w.g, I, e->W(o.w2), I, ws + w.s2, H, nullptr, nullptr, P + o.b2,
                ws + w.y1, H, e->key(XS_LAYER0 + 8 * l + 3, pd), 1, 0, st));
        CK(ln_forward(dt, ws + w.s2, P + o.fflnw, P + o.fflnb, c.layer_norm_eps, ws + e->ws_x[l + 1], (float*)(ws + w.st2),
                      (float*)(ws + w.st2) + T, T, H, kNoDrop, st, pf2));
    }
    if (!last) return MB_OK;
    // final dropout (xlnet.py:396) on the only row SequenceSummary("last") reads, then summary -> tanh -> dropout -> logits_proj
    CK(last_token_forward(dt, ws + e->ws_x[c.n_layer], ws + e->ws_xs, B, L, H, e->key(XS_FINAL, pd), st));
    float* z = (float*)(ws + e->ws_head_z);
    CK(gemm(dt, GEMM_NT, EPI_BIAS_F32, B, H, H, ws + e->ws_xs, H, e->W(e->wsum), H, nullptr, H, nullptr, z, P + e->bsum, nullptr, 0,
            kNoDrop, 1, 64, st));
    if (loss && !(e->in_step && e->loss_cleared)) CK(zero_fill(loss, 4, st));
    CK(head_forward(z, P + e->wc, P + e->bc, labels, (float*)(ws + e->ws_head_pooled), logits, loss, loss_run, B, H, c.num_labels,
                    e->key(XS_HEAD, c.summary_last_dropout), st));
    return MB_OK;
}

int mb_xlnet_forward(mb_xlnet_engine* e, const int64_t* input_ids, const float* visual, const float* acoustic,
                     const int64_t* attention_mask, const int64_t* token_type_ids, const float* labels, int B, int L,
                     int training, uint64_t seed, uint64_t step, float* logits, float* loss, float* loss_run, void* stream) {
    return xl_forward_range(e, input_ids, visual, acoustic, attention_mask, token_type_ids, labels, B, L, training, seed, step, logits, loss,
                            loss_run, stream, 0, e->c.n_layer, true, true);
}

int mb_xlnet_backward(mb_xlnet_engine* e, const float* dlogits, const float* labels, float loss_scale, int stage_begin,
                      int stage_end, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const mb_xlnet_config& c = e->c;
    if (!e->G || !e->ran_forward) return MB_ERR_ARG;
    const int dt = c.dtype, H = c.d_model, I = c.d_inner, B = e->B, L = e->L, T = B * L, nh = c.n_head, NL = c.n_layer;
    const int Tk = (int)align_up((size_t)T, 64), Rk = (int)align_up((size_t)B * 2 * L, 64);
    const size_t es = esize(dt);
    if (stage_begin < 0) stage_begin = 0;
    if (stage_end > NL + 2) stage_end = NL + 2;
    if (stage_begin == 0) CK(e->begin_backward_pass(e->G, st));
    float* P = e->P; float* G = e->G;
    char* ws = e->ws;
    const float pd = c.dropout;
    const bool hd = e->training && pd > 0.f;
    const GradAcc acc = e->acc_of(ws, G);          // deterministic mode (MB_DETERMINISTIC=1): where the multi-writer sums go (as engine.hip)
    // Riders (kernels.h AdamRide, xl_enqueue_step): up to `budget` parameters from the TOP of what is final while layer l's backward runs -- the
    // GEMM weights of layers l+1 .. NL-1 (q | k | v | o | r | layer_1 | layer_2, contiguous per layer; r's second k half has been added by then)
    // minus what earlier launches took -- as `blocks` extra workgroups of a launch.  The sweep at the end is [0, ride_cursor) + the rest.
    auto take_ride = [&](int l, size_t budget, int blocks) -> AdamRide {
        AdamRide r = {};
        if (!e->ride_m || !e->ride_v || l + 1 >= NL || blocks < 8 || e->ride_cursor <= e->lo[l + 1].q) return r;
        const size_t take = std::min(e->ride_cursor - e->lo[l + 1].q, budget) / 1024 * 1024;
        const size_t re = e->ride_cursor, rb = re - take;
        const bool sh_ok = dt != DT_BF16 || (e->sh_begin <= rb && re <= e->sh_end);
        if (take == 0 || rb % 4 || !sh_ok) return r;
        const bool keep = e->keep_in_step() && e->stale_begin <= rb && re <= e->stale_end;
        r = AdamRide{P + rb, G + rb, e->ride_m + rb, e->ride_v + rb, dt == DT_BF16 ? (bf16*)(e->SH + rb * 2) : nullptr, take / 4, e->adam_state(ws),
                     blocks / 8 * 8, keep ? 0 : 1};
        e->ride_cursor = rb;
        return r;
    };
    for (int stage = stage_begin; stage < stage_end; ++stage) {
        if (stage == 0) {
            CK(head_backward(dt, dlogits, e->logits, labels, loss_scale, (const float*)(ws + e->ws_head_pooled), P + e->wc,
                             ws + e->ws_dz, G + e->wc, G + e->bc, B, H, c.num_labels, e->key(XS_HEAD, c.summary_last_dropout), st,
                             acc, G + e->bsum, ws + e->ws_dxa, (size_t)T * H * es));     // + summary bias gradient, + dx cleared
            CK(gemm(dt, GEMM_TN, EPI_ACCUM_F32, H, H, B, ws + e->ws_dz, H, ws + e->ws_xs, H, nullptr, H, nullptr, G + e->wsum, nullptr,
                    nullptr, 0, kNoDrop, 1, 64, st));
            CK(gemm(dt, GEMM_NN, EPI_ADD_RES, B, H, H, ws + e->ws_dz, H, e->W(e->wsum), H, ws + e->ws_dxs, H, nullptr, nullptr, nullptr,
                    nullptr, 0, kNoDrop, 1, 64, st));
            CK(last_token_backward(dt, ws + e->ws_dxs, ws + e->ws_dxa, B, L, H, e->key(XS_FINAL, pd), st));
        } else if (stage <= NL) {
            const int l = NL - stage;
            const XlLayerOff& o = e->lo[l];
            const XlLayerWs& w = e->lw[l];
            const char* xin = (l == c.injection_index) ? ws + e->ws_magout : ws + e->ws_x[l];
            char* dx = ws + e->ws_dxa;
            char* t1 = ws + e->ws_dxb;
            const int par = l & 1;
            char* dsA = ws + e->ws_dsa[par];
            char* dzdA = hd ? ws + e->ws_dzda[par] : dsA;
            char* dsB = ws + e->ws_dsb[par];
            char* dzdB = hd ? ws + e->ws_dzdb[par] : dsB;
            char* du = ws + e->ws_du[par];
            char* dkr = ws + e->ws_dkr[par];
            int nblk = 0;
            float* lnp_a = (float*)(ws + e->ws_lnp_a) + (size_t)l * e->lnp_stride;
            float* lnp_b = (float*)(ws + e->ws_lnp_b) + (size_t)l * e->lnp_stride;
            const bool defer_ln = e->in_step && NL + 1 <= MB_LN_MAX_LAYERS;      // as in engine.hip: one reduction launch for all layers
            const bool mag_slabs = defer_ln && c.injection_index >= 1 && c.injection_index < NL;     // MAG's backward runs before that launch
            // ---- feed-forward block
            CK(ln_backward_partials(dt, dx, ws + w.s2, P + o.fflnw, (const float*)(ws + w.st2), (const float*)(ws + w.st2) + T, dsA,
                                    hd ? dzdA : nullptr, lnp_a, &nblk, T, H, e->key(XS_LAYER0 + 8 * l + 3, pd), st,
                                    Prefetch{e->prefetch ? e->W(o.w1) : nullptr, (size_t)2 * I * H * (dt == DT_BF16 ? 2 : 4), nullptr}));
            // the layer's seven weight gradients go out as ONE grouped launch once every dY exists (MB_GROUP_WGRAD=0: one by one)
            char* dqkv = ws + e->ws_dqkv[par];
            // The relative-position problem d r = pos^T dkr reduces over B * 2L rows -- twice the K of the other six -- so in a launch
            // that is ONE round of tiles its 36 tiles ran 1.6 x as long as everybody else's and set the kernel's duration (71 us against
            // 48 us for MAG-BERT's group with 16 % less work).  Its K range is cut in two: the second half is an eighth problem that
            // STORES into a scratch [H][H], added to the gradient by a 3 us launch behind the group (MB_XL_SPLIT_R=0: one problem).
            const int Rk1 = (e->split_r && Rk >= 256) ? (int)align_up((size_t)Rk / 2, 64) : Rk;
            GemmArgs wg[8] = {wgrad_args(H, I, Tk, dzdA, H, ws + w.g, I, G + o.w2, I),
                              wgrad_args(I, H, Tk, du, I, ws + w.y1, H, G + o.w1, H),
                              wgrad_args(H, H, Tk, dzdB, H, ws + w.vec, H, G + o.o, H),                 // d o[h][nd] = dzd^T vec
                              wgrad_args(H, H, Rk1, ws + e->ws_pos, H, dkr, H, G + o.r, H),   // d r = pos^T dkr, rows [0, Rk1)
                              wgrad_args(H, H, Tk, xin, H, dqkv, 3 * H, G + o.q, H),
                              wgrad_args(H, H, Tk, xin, H, dqkv + (size_t)H * es, 3 * H, G + o.k, H),
                              wgrad_args(H, H, Tk, xin, H, dqkv + (size_t)2 * H * es, 3 * H, G + o.v, H),
                              wgrad_args(H, H, Rk - Rk1, ws + e->ws_pos + (size_t)Rk1 * H * es, H, dkr + (size_t)Rk1 * H * es, H,
                                         (float*)(ws + e->ws_rhalf), H)};                       // ... rows [Rk1, Rk) -> scratch
            const int nwg = Rk1 < Rk ? 8 : 7;
            int wtile = e->group_wgrad;
            if (wtile == 256 && !gemm_grouped_tn_ok(dt, wg, nwg, 256)) wtile = 128;      // (a short k range in the group: the 128 x 128 kernel)
            const bool grouped = wtile > 0 && gemm_grouped_tn_ok(dt, wg, nwg, wtile);
            if (grouped) {
                for (GemmArgs& a : wg) a.overwrite = e->ow_pass ? 1 : 0;
                wg[7].overwrite = 1;
            }
            if (!grouped) CK(wgrad(dt, H, I, Tk, dzdA, H, ws + w.g, I, G + o.w2, I, st));
            const bool riding = grouped && !e->deferred && e->ride_m != nullptr;
            auto take_l = [&](size_t budget, int blocks) { return take_ride(l, budget, blocks); };
            CK(dgrad_with_riders(dt, EPI_DGELU, T, I, H, dzdA, H, e->W(o.w2), I, du, I, ws + w.u, I, G + o.b1, e->key(XS_LAYER0 + 8 * l + 2, pd), acc, st,
                                 riding, e->ride_opts, e->cu_count(), take_l));
            if (!grouped) CK(wgrad(dt, I, H, Tk, du, I, ws + w.y1, H, G + o.w1, H, st));
            CK(dgrad_with_riders(dt, EPI_ADD_RES, T, H, I, du, I, e->W(o.w1), H, t1, H, dsA, H, nullptr, kNoDrop, GradAcc{}, st, riding, e->ride_opts,
                                 e->cu_count(), take_l));
            // ---- relative attention block
            CK(ln_backward_partials(dt, t1, ws + w.s1, P + o.ralnw, (const float*)(ws + w.st1), (const float*)(ws + w.st1) + T, dsB,
                                    hd ? dzdB : nullptr, lnp_b, &nblk, T, H, e->key(XS_LAYER0 + 8 * l + 1, pd), st,
                                    Prefetch{e->prefetch ? e->W(o.q) : nullptr, (size_t)5 * H * H * (dt == DT_BF16 ? 2 : 4), nullptr}));
            if (!defer_ln) {
                float* const dst6[6] = {G + o.fflnw, G + o.fflnb, G + o.b2, G + o.ralnw, G + o.ralnb, nullptr};
                CK(ln_reduce_partials(lnp_a, lnp_b, nblk, H, dst6, st, acc));
            } else if (l == 0) {
                LnReduceDst dst = {};
                for (int k = 0; k < NL; ++k) {
                    const XlLayerOff& ok = e->lo[k];
                    float* const d6[6] = {G + ok.fflnw, G + ok.fflnb, G + ok.b2, G + ok.ralnw, G + ok.ralnb, nullptr};
                    for (int q = 0; q < 6; ++q) dst.d[k][q] = d6[q];
                }
                if (mag_slabs) {         // the six column sums of MAG's gate (mag_bwd_impl below wrote slot NL in the injection layer's stage)
                    float* const m6[6] = {G + e->mag_bhv, G + e->mag_bha, G + e->mag_bv, G + e->mag_ba, G + e->mag_lnw, G + e->mag_lnb};
                    for (int q = 0; q < 6; ++q) dst.d[NL][q] = m6[q];
                    dst.nblk[NL] = e->mag_nblk;
                }
                CK(ln_reduce_partials_layers((const float*)(ws + e->ws_lnp_a), (const float*)(ws + e->ws_lnp_b), e->lnp_stride,
                                             mag_slabs ? NL + 1 : NL, nblk, H, dst, st, acc));
            }
            if (!grouped) CK(wgrad(dt, H, H, Tk, dzdB, H, ws + w.vec, H, G + o.o, H, st));
            CK(dgrad_with_riders(dt, EPI_ADD_RES, T, H, H, dzdB, H, e->W(o.o), H, ws + e->ws_dvec, H, nullptr, 0, nullptr, kNoDrop, GradAcc{}, st, riding,
                                 e->ride_opts, e->cu_count(), take_l));
            // (riders of the two relative-attention backward launches: 576 workgroups in 768 slots each at L = 50, latency-bound hosts)
            AdamRide rq = {}, rkv = {};
            if (riding && e->ride_attn) {
                int free_slots = xlnet_attention_backward_free_slots(dt, L, B * nh, e->cu_count());
                if (e->ride_attn_blocks > 0 && free_slots > 0) free_slots = e->ride_attn_blocks;
                const int blocks = std::min(free_slots, 2 * e->cu_count()) / 8 * 8;
                if (blocks >= 8) {
                    // same box, B = 48 L = 50: 4.253 ms without riders | 4.212 dgrad hosts only | 4.155 at 1.25 M + 0.83 M in the two attention launches |
                    // 4.129 at 1.8 M + 1.2 M (this: 750 per token) | 4.128 at 2.4 M + 1.6 M   (profiles/r06_xlnet_riders.txt)
                    // (later in the round, with the 128 x 64 tile's smaller dgrad riders: 4.040 ms at 1.8 M | 4.021 at 2.4 M | 4.011 at 3 M -- this:
                    //  1,250 per token; profiles/r06_final_defaults.txt; and further: 4.000 at 3 M | 3.983 at 4.2 M | 3.965 at 5.4 M, where the two
                    //  hosts take everything that is final -- this: 2,250 per token; profiles/r06_xlnet_ride_budget2.txt)
                    const size_t budget = e->ride_attn_params > 0 ? (size_t)e->ride_attn_params : (size_t)2250 * (size_t)T;
                    rq = take_ride(l, budget / 1024 * 1024, blocks);
                    rkv = take_ride(l, (budget * 2 / 3) / 1024 * 1024, blocks);
                }
            }
            CK(xlnet_attention_backward(dt, ws + w.qkv, ws + w.kr, P + o.rwb, P + o.rrb, P + o.rsb, P + o.seg, e->seg, e->mask,
                                        ws + w.psave, ws + e->ws_dvec, ws + e->ws_gsave, dqkv, dkr, G + o.rwb, G + o.rrb,
                                        G + o.rsb, G + o.seg, B, L, nh, e->key(XS_LAYER0 + 8 * l + 0, pd), st,
                                        e->head_mask ? e->head_mask + (size_t)l * nh : nullptr, acc, rq.blocks ? &rq : nullptr,
                                        rkv.blocks ? &rkv : nullptr));
            if (grouped && e->deferred) {
                if (!e->side) {
                    CK((int)hipStreamCreateWithFlags(&e->side, hipStreamNonBlocking));
                    e->evs.assign((size_t)2 * NL, nullptr);
                    for (auto& ev : e->evs) CK((int)hipEventCreateWithFlags(&ev, hipEventDisableTiming));
                }
                CK((int)hipEventRecord(e->evs[2 * l], st));                       // every dY of the layer is final on `st`
                CK((int)hipStreamWaitEvent(e->side, e->evs[2 * l], 0));
                CK(gemm_grouped_tn_launch(dt, wg, nwg, wtile, e->side));
                if (nwg == 8) CK(add_f32(G + o.r, (const float*)(ws + e->ws_rhalf), (size_t)H * H, e->side));
                CK((int)hipEventRecord(e->evs[2 * l + 1], e->side));              // "weight gradients of layer l are final"
            } else if (grouped) {
                CK(e->prof_mark(2 * l, st));
                CK(gemm_grouped_tn_launch(dt, wg, nwg, wtile, st));
                CK(e->prof_mark(2 * l + 1, st));
                if (nwg == 8) CK(add_f32(G + o.r, (const float*)(ws + e->ws_rhalf), (size_t)H * H, st));
            } else {
                CK(wgrad(dt, H, H, Rk, ws + e->ws_pos, H, dkr, H, G + o.r, H, st));        // (one by one: the whole K range, no scratch)
                CK(wgrad(dt, H, H, Tk, xin, H, dqkv, 3 * H, G + o.q, H, st));
                CK(wgrad(dt, H, H, Tk, xin, H, dqkv + (size_t)H * es, 3 * H, G + o.k, H, st));
                CK(wgrad(dt, H, H, Tk, xin, H, dqkv + (size_t)2 * H * es, 3 * H, G + o.v, H, st));
            }
            // dx_in = dq Wq^T + dk Wk^T + dv Wv^T + dsB   (W stored [h_in][nd] = the row operand of an NT GEMM)
            char* t2 = ws + e->ws_dvec;
            if (e->fuse_qkv) {
                // ONE GEMM, K = 3 H: A = dqkv as it lies, B = [Wq | Wk | Wv] along k (a row-major B whose k is cut into three segments)
                CK(gemm(dt, GEMM_NT, EPI_ADD_RES, T, H, 3 * H, dqkv, 3 * H, e->W(o.q), H, dx, H, nullptr, nullptr, nullptr, dsB, H, kNoDrop, 1, 0,
                        st, H, o.k - o.q));
            } else {
                CK(gemm(dt, GEMM_NT, EPI_ADD_RES, T, H, H, dqkv, 3 * H, e->W(o.q), H, t1, H, nullptr, nullptr, nullptr, dsB, H, kNoDrop, 1, 0, st));
                CK(gemm(dt, GEMM_NT, EPI_ADD_RES, T, H, H, dqkv + (size_t)H * es, 3 * H, e->W(o.k), H, t2, H, nullptr, nullptr, nullptr, t1, H,
                        kNoDrop, 1, 0, st));
                CK(gemm(dt, GEMM_NT, EPI_ADD_RES, T, H, H, dqkv + (size_t)2 * H * es, 3 * H, e->W(o.v), H, dx, H, nullptr, nullptr, nullptr, t2,
                        H, kNoDrop, 1, 0, st));
            }
            if (e->mems) {
                // training with cached memories (xlnet.py:81-91: cache_mem detaches): rows [0, mlen) of this layer's input were REPLACED by
                // mems[l] in the forward, so neither the memory (detached) nor what the layer below computed for those rows (overwritten)
                // receives a gradient -- the rows are cleared at every seam.  What the rows contributed as keys / values to THIS layer's
                // k / v weight gradients stays (the reference's k_head_h = einsum(cat([mems, h]), k) does the same).
                CK((int)hipMemset2DAsync(dx, (size_t)L * H * es, 0, (size_t)e->mlen * H * es, (size_t)B, st));
            }
            if (l == c.injection_index) {      // MAG sits in front of this layer
                int mblk = 0;
                CK(mag_bwd_impl(dt, dx, ws + e->ws_x[l], P + e->mag_bhv, P + e->mag_bha, P + e->mag_bv, P + e->mag_ba,
                                P + e->mag_lnw, c.beta_shift, e->key(XS_MAG, c.mag_dropout), ws + e->ws_mag, e->mw, t1, nullptr,
                                nullptr, G + e->mag_whv, G + e->mag_bhv, G + e->mag_wha, G + e->mag_bha, G + e->mag_wv, G + e->mag_bv,
                                G + e->mag_wa, G + e->mag_ba, G + e->mag_lnw, G + e->mag_lnb, T, H, c.visual_dim, c.acoustic_dim, true,
                                st, acc, true, (float*)(ws + e->ws_lnp_a) + (size_t)NL * e->lnp_stride,
                                (float*)(ws + e->ws_lnp_b) + (size_t)NL * e->lnp_stride, &mblk, e->ow_pass));
                e->mag_nblk = mblk;
                if (!mag_slabs) {        // not a single-call step (or MAG in front of layer 0): reduce MAG's slabs right away
                    float* const m6[6] = {G + e->mag_bhv, G + e->mag_bha, G + e->mag_bv, G + e->mag_ba, G + e->mag_lnw, G + e->mag_lnb};
                    CK(ln_reduce_partials((const float*)(ws + e->ws_lnp_a) + (size_t)NL * e->lnp_stride,
                                          (const float*)(ws + e->ws_lnp_b) + (size_t)NL * e->lnp_stride, mblk, H, m6, st, acc));
                }
                {   // dx <- t1 as a kernel (a captured step holds kernel nodes only)
                    PrologueArgs cp = {};
                    cp.src[0] = (const uint32_t*)t1; cp.dst[0] = (uint32_t*)dx; cp.dwords[0] = (uint32_t)((size_t)T * H * es / 4); cp.ncopies = 1;
                    CK(step_prologue(cp, st));
                }
            }
            // deferred join: main waits for layer l+1's launch only now (its dY buffers have the parity of layer l-1, written next)
            if (grouped && e->deferred && l + 1 < NL) CK((int)hipStreamWaitEvent(st, e->evs[2 * (l + 1) + 1], 0));
        } else {
            if (e->deferred && e->side) CK((int)hipStreamWaitEvent(st, e->evs[1], 0));       // weight gradients of layer 0
            CK(gather_drop_backward(dt, ws + e->ws_dxa, e->ids, e->ids ? G + e->word : (float*)(ws + e->ws_demb), T, H, e->key(XS_EMB, pd), st, acc));
            // deterministic mode: the integer sums become part of the fp32 gradients before anybody (AdamW, an exchange) reads them
            CK(grad_fold(acc, G, e->det_begin, e->det_end, st));
        }
    }
    return MB_OK;
}

// ------------------------------------------------------------------------------------------------ whole step (as mb_bert_train_step)
static int xl_adamw_decay_range(mb_xlnet_engine* e, float* m, float* v, size_t b, size_t en, hipStream_t st);
static int xl_enqueue_step(mb_xlnet_engine* e, int B, int L, float* logits, float* loss, float* loss_run, float* m, float* v,
                           float loss_scale, hipStream_t st) {
    char* ws = e->ws;
    const float* lab = (const float*)(ws + e->ws_in_lab);
    CK(mb_xlnet_forward(e, (const int64_t*)(ws + e->ws_in_ids), (const float*)(ws + e->ws_in_vis), (const float*)(ws + e->ws_in_aco),
                        (const int64_t*)(ws + e->ws_in_mask), (const int64_t*)(ws + e->ws_in_seg), lab, B, L, 1, 0, 0, logits, loss,
                        loss_run, st));
    // riders (MB_ADAMW_RIDE): layers 1 .. NL-1 are updated inside launches of the backward of layers 0 .. NL-2 (mb_xlnet_backward: take_ride);
    // whether a launch really carried one is decided there, so the sweep below asks the engine what is still to do
    const bool ride = e->adam_ride && m && v && e->c.dtype == DT_BF16 && e->group_wgrad > 0 && !e->deferred && e->c.n_layer > 1 && e->lo[0].q == 0 &&
                      !e->prof && !e->mems;
    e->ride_m = ride ? m : nullptr; e->ride_v = ride ? v : nullptr;
    e->ride_cursor = e->wsum;
    const int rb = mb_xlnet_backward(e, nullptr, lab, loss_scale, 0, e->c.n_layer + 2, st);
    e->ride_m = e->ride_v = nullptr;
    CK(rb);
    if (m && v && ride) {
        const AdamArgs none = {};
        const size_t nd = e->n_decay, n = e->n_trainable;
        CK(e->prof_mark(2 * e->c.n_layer, st));
        CK(xl_adamw_decay_range(e, m, v, 0, e->ride_cursor, st));          // what no launch carried (layer 0 always)
        CK(xl_adamw_decay_range(e, m, v, e->wsum, nd, st));
        CK(adamw_step(e->P + nd, e->G + nd, m + nd, v + nd, nullptr, n - nd, 0, 0, 0, none, 1, st, e->adam_state(ws) + 1));
        CK(e->prof_mark(2 * e->c.n_layer + 1, st));
    } else if (m && v) {
        const AdamArgs none = {};
        const size_t nd = e->n_decay, n = e->n_trainable;        // the frozen mask_emb slot behind n_trainable is never updated
        void* sh = e->c.dtype == DT_BF16 ? (void*)e->SH : nullptr;
        const bool keep = e->keep_in_step();          // the layers' GEMM weight gradients: overwritten by the next backward, not zeroed
        CK(e->prof_mark(2 * e->c.n_layer, st));
        CK(adamw_step(e->P, e->G, m, v, sh, nd, nd, e->sh_begin, e->sh_end, none, 1, st, e->adam_state(ws), keep ? e->stale_begin : 0,
                      keep ? e->stale_end : 0));
        CK(adamw_step(e->P + nd, e->G + nd, m + nd, v + nd, nullptr, n - nd, 0, 0, 0, none, 1, st, e->adam_state(ws) + 1));
        CK(e->prof_mark(2 * e->c.n_layer + 1, st));
    }
    return MB_OK;
}

int mb_xlnet_train_step(mb_xlnet_engine* e, const int64_t* input_ids, const float* visual, const float* acoustic,
                        const int64_t* attention_mask, const int64_t* token_type_ids, const float* labels, int B, int L,
                        uint64_t seed, uint64_t step, float* logits, float* loss, float* loss_run, float* m, float* v, float lr,
                        float beta1, float beta2, float eps, float weight_decay, int opt_step, int correct_bias, float grad_scale,
                        float loss_scale, int mode, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (!e || !e->P || !e->G || !e->ws) return MB_ERR_ARG;
    const mb_xlnet_config& c = e->c;
    if (B < 1 || B > c.max_batch || L < 1 || L > c.max_seq) return MB_ERR_SHAPE;
    if (!input_ids || !visual || !acoustic || !attention_mask || !token_type_ids || !labels || !logits || !loss) return MB_ERR_ARG;
    if ((m == nullptr) != (v == nullptr) || (mode != 1 && mode != 2)) return MB_ERR_ARG;
    if (e->deferred) return MB_ERR_MODE;          // MB_OVERLAP_WGRAD=1: the side-stream scheme is driven stage by stage (mb_xlnet_backward)
    if (e->head_mask || e->emb_in || e->perm || e->mems) return MB_ERR_MODE;      // head_mask / inputs_embeds / perm_mask / mems are arguments of explicit forwards only
    e->training = 1;
    CK(xl_prepare_pass(e, B * L, st));
    return train_step_impl(e, e->ws, c.visual_dim, c.acoustic_dim, c.num_labels, input_ids, visual, acoustic, attention_mask, token_type_ids,
                           labels, B, L, seed, step, logits, loss, loss_run, m, v, lr, beta1, beta2, eps, weight_decay, opt_step,
                           correct_bias, grad_scale, loss_scale, mode, e->prof, st,
                           [&](int, float* lg, float* ls, float* lr_, float* m_, float* v_, float sc, hipStream_t s) {
                               return xl_enqueue_step(e, B, L, lg, ls, lr_, m_, v_, sc, s);
                           });
}

// ---- data-parallel step in one call (as mb_bert_train_step_dp: include/magbert_hip.h, csrc/comm.hip).  Segments: [0, nb) = (segment
// 0: forward + head) + the backward of plan[s] layers (+ the last: the embedding stage) | nb = AdamW over the GEMM weights of the layers
// reduced early | nb + 1 = AdamW over the last segment's layers and everything else
static int xl_adamw_decay_range(mb_xlnet_engine* e, float* m, float* v, size_t b, size_t en, hipStream_t st) {
    if (en <= b) return MB_OK;
    const AdamArgs none = {};
    const bool keep = e->keep_in_step();
    auto clampr = [&](size_t x) { return x < b ? (size_t)0 : (x > en ? en - b : x - b); };
    void* sh = e->c.dtype == DT_BF16 ? (void*)(e->SH + b * 2) : nullptr;
    return adamw_step(e->P + b, e->G + b, m + b, v + b, sh, en - b, en - b, clampr(e->sh_begin), clampr(e->sh_end), none, 1, st,
                      e->adam_state(e->ws), keep ? clampr(e->stale_begin) : 0, keep ? clampr(e->stale_end) : 0);
}
// sharded update: of every chunk of layer GEMM weights inside [b, en) only this rank's slice + the replicated remainder (engine.hip)
static int xl_adamw_decay_range_dp(mb_xlnet_engine* e, const mb_comm* comm, const DpSpec& sp, float* m, float* v, size_t b, size_t en, hipStream_t st) {
    if (!comm->shard) return xl_adamw_decay_range(e, m, v, b, en, st);
    std::vector<std::pair<size_t, size_t>> ch(sp.chunk.begin(), sp.chunk.begin() + sp.n_sharded);      // (the rest is replicated)
    std::sort(ch.begin(), ch.end());
    size_t cur = b;
    ZeroRanges dead = {};
    for (const auto& c : ch) {
        if (c.second <= cur || c.first >= en) continue;
        if (c.first < cur || c.second > en) return MB_ERR_MODE;
        CK(xl_adamw_decay_range(e, m, v, cur, c.first, st));
        const ShardSlice sl = dp_shard_slice(comm, c.first, c.second);
        CK(xl_adamw_decay_range(e, m, v, sl.mine_b, sl.mine_e, st));
        CK(xl_adamw_decay_range(e, m, v, sl.rem_b, sl.rem_e, st));
        if (!e->keep_in_step()) {
            if (dead.n + 2 > MB_ZERO_MAX) { CK(zero_fill_ranges(dead, st)); dead = ZeroRanges{}; }       // (add() drops what does not fit)
            dead.add(e->G + c.first, (sl.mine_b - c.first) * 4);
            dead.add(e->G + sl.mine_e, (sl.rem_b - sl.mine_e) * 4);
        }
        cur = c.second;
    }
    if (dead.n) CK(zero_fill_ranges(dead, st));
    return xl_adamw_decay_range(e, m, v, cur, en, st);
}
static int xl_enqueue_step_dp(mb_xlnet_engine* e, int seg, const std::vector<int>& plan, const mb_comm* comm, const DpSpec& sp, int B, int L,
                              float* logits, float* loss, float* loss_run, float* m, float* v, float loss_scale, hipStream_t st) {
    char* ws = e->ws;
    const int NL = e->c.n_layer, nb = (int)plan.size();
    const float* lab = (const float*)(ws + e->ws_in_lab);
    const int nf = comm->nf;          // sharded update with several pieces: forward-only segments in front (engine.hip)
    auto layers_of = [&](int chunk, int& l0, int& l1) { l1 = NL; for (int s = 0; s < chunk; ++s) l1 -= plan[s]; l0 = l1 - plan[chunk]; };
    if (seg <= nf) {
        int l0 = 0, l1 = NL;
        if (nf > 0) layers_of(nb - 1 - seg, l0, l1);
        CK(xl_forward_range(e, (const int64_t*)(ws + e->ws_in_ids), (const float*)(ws + e->ws_in_vis), (const float*)(ws + e->ws_in_aco),
                            (const int64_t*)(ws + e->ws_in_mask), (const int64_t*)(ws + e->ws_in_seg), lab, B, L, 1, 0, 0, logits, loss,
                            loss_run, st, l0, l1, seg == 0, seg == nf));
        if (seg < nf) return MB_OK;
    }
    seg -= nf;
    if (seg < nb) {
        int done = 0;
        for (int s = 0; s < seg; ++s) done += plan[s];
        return mb_xlnet_backward(e, nullptr, lab, loss_scale, seg == 0 ? 0 : 1 + done, seg == nb - 1 ? NL + 2 : 1 + done + plan[seg], st);
    }
    const AdamArgs none = {};
    const size_t nd = e->n_decay, n = e->n_trainable;
    const size_t split = e->lo[plan[nb - 1] < NL ? plan[nb - 1] : 0].q;
    if (seg == nb) {
        CK(e->prof_mark(2 * NL, st));
        if (nb > 1) return xl_adamw_decay_range_dp(e, comm, sp, m, v, split, e->wsum, st);
        // (one backward segment: no early range -- dp_between waited for everything -- so this segment takes the no-decay slab)
        return adamw_step(e->P + nd, e->G + nd, m + nd, v + nd, nullptr, n - nd, 0, 0, 0, none, 1, st, e->adam_state(ws) + 1);
    }
    CK(xl_adamw_decay_range_dp(e, comm, sp, m, v, 0, nb > 1 ? split : e->wsum, st));
    CK(xl_adamw_decay_range(e, m, v, e->wsum, nd, st));
    if (nb > 1) CK(adamw_step(e->P + nd, e->G + nd, m + nd, v + nd, nullptr, n - nd, 0, 0, 0, none, 1, st, e->adam_state(ws) + 1));
    return e->prof_mark(2 * NL + 1, st);
}

int mb_xlnet_train_step_dp(mb_xlnet_engine* e, const int64_t* input_ids, const float* visual, const float* acoustic,
                           const int64_t* attention_mask, const int64_t* token_type_ids, const float* labels, int B, int L,
                           uint64_t seed, uint64_t step, float* logits, float* loss, float* loss_run, float* m, float* v, float lr,
                           float beta1, float beta2, float eps, float weight_decay, int opt_step, int correct_bias, float grad_scale,
                           float loss_scale, int mode, void* stream, mb_comm* comm) {
    hipStream_t st = (hipStream_t)stream;
    if (!e || !e->P || !e->G || !e->ws || !comm) return MB_ERR_ARG;
    const mb_xlnet_config& c = e->c;
    if (B < 1 || B > c.max_batch || L < 1 || L > c.max_seq) return MB_ERR_SHAPE;
    if (!input_ids || !visual || !acoustic || !attention_mask || !token_type_ids || !labels || !logits || !loss) return MB_ERR_ARG;
    if (!m || !v || (mode != 1 && mode != 2)) return MB_ERR_ARG;
    if (e->deferred || e->head_mask || e->emb_in || e->perm || e->mems) return MB_ERR_MODE;
    const int NL = c.n_layer;
    const std::vector<int> plan = dp_chunk_plan(NL);
    const int nb = (int)plan.size();
    DpSpec sp;
    for (int s = 0, hi = NL; s < nb; ++s) {
        const int lo_l = hi - plan[s];
        sp.chunk.push_back({e->lo[lo_l].q, hi < NL ? e->lo[hi].q : e->wsum});
        hi = lo_l;
    }
    sp.tail_begin = e->wsum; sp.tail_end = e->n_trainable;
    sp.word_off = e->word; sp.word_rows = c.vocab_size; sp.H = c.d_model;
    sp.ids = (const int64_t*)(e->ws + e->ws_in_ids); sp.T = B * L;
    sp.n_sharded = dp_sharded_chunks(comm, nb);
    const int nf = dp_forward_segments(comm, nb);
    comm->nf = nf;
    if (comm->shard) {
        const bool bf = c.dtype == DT_BF16;
        for (const auto& ch : sp.chunk)
            if (bf && (!e->SH || ch.first < e->sh_begin || ch.second > e->sh_end)) return MB_ERR_MODE;
        sp.gather_base = bf ? (char*)e->SH : (char*)e->P; sp.gather_es = bf ? 2 : 4;
    }
    CK(dp_step_begin(comm, st, nf > 0));          // (sharded update: the previous step's all-gathers -- cut mode: awaited piece by piece)
    e->training = 1;
    CK(xl_prepare_pass(e, B * L, st));
    int variant = 1;
    for (int x : plan) variant = variant * 13 + x;
    variant = (variant * 4 + comm->event_mode) * 2 + (comm->shard ? 1 : 0);
    return train_step_impl(e, e->ws, c.visual_dim, c.acoustic_dim, c.num_labels, input_ids, visual, acoustic, attention_mask, token_type_ids,
                           labels, B, L, seed, step, logits, loss, loss_run, m, v, lr, beta1, beta2, eps, weight_decay, opt_step,
                           correct_bias, grad_scale, loss_scale, mode, e->prof, st,
                           [&](int sg, float* lg, float* ls, float* lr_, float* m_, float* v_, float sc, hipStream_t s) {
                               CK(dp_segment_begin(comm, nb, sg, s));
                               CK(xl_enqueue_step_dp(e, sg, plan, comm, sp, B, L, lg, ls, lr_, m_, v_, sc, s));
                               CK(dp_segment_end(comm, nb, sg, s));
                               return (int)MB_OK;
                           },
                           nf + nb + 2, [&](int sg, hipStream_t s) { return dp_between(comm, sp, e->G, sg, s); }, variant, comm, dp_finish_segment_graph);
}

int mb_xlnet_set_perm_mask(mb_xlnet_engine* e, const uint8_t* perm) {
    if (!e) return MB_ERR_ARG;
    e->perm = perm;
    return MB_OK;
}
int mb_xlnet_set_mems(mb_xlnet_engine* e, const void* mems, int mlen) {
    if (!e || (mems && (mlen < 1 || mlen >= e->c.max_seq))) return MB_ERR_ARG;
    e->mems = (const char*)mems; e->mlen = mems ? mlen : 0;
    return MB_OK;
}
int mb_xlnet_set_head_mask(mb_xlnet_engine* e, const float* head_mask) {
    if (!e) return MB_ERR_ARG;
    e->head_mask = head_mask;
    return MB_OK;
}
int mb_xlnet_set_profiling(mb_xlnet_engine* e, int on) {
    if (!e) return MB_ERR_ARG;
    return e->set_profiling(on, e->c.n_layer);
}
int mb_xlnet_profile_wgrad_us(mb_xlnet_engine* e, float* avg_us) {
    if (!e || !e->ow_covers || e->deferred) return MB_ERR_ARG;
    return e->prof_span_us(0, e->c.n_layer, avg_us);
}
int mb_xlnet_profile_adamw_us(mb_xlnet_engine* e, float* us) {
    if (!e) return MB_ERR_ARG;
    return e->prof_span_us(2 * e->c.n_layer, 1, us);
}
int mb_xlnet_materialize_grads(mb_xlnet_engine* e, void* stream) {
    if (!e) return MB_ERR_ARG;
    return e->materialize_grads(e->G, (hipStream_t)stream);
}
int mb_xlnet_grads_stale(const mb_xlnet_engine* e) { return e && e->grads_stale ? 1 : 0; }
int mb_xlnet_mark_grads_zero(mb_xlnet_engine* e, int known_zero) {
    if (!e) return MB_ERR_ARG;
    e->grads_zero = known_zero != 0;
    if (known_zero) e->grads_stale = false;
    return MB_OK;
}

int mb_xlnet_load_batch(mb_xlnet_engine* e, const int64_t* input_ids, const float* visual, const float* acoustic,
                        const int64_t* attention_mask, const int64_t* token_type_ids, const float* labels, int B, int L,
                        const void** staged6, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (!e || !e->ws || !staged6) return MB_ERR_ARG;
    const mb_xlnet_config& c = e->c;
    if (B < 1 || B > c.max_batch || L < 1 || L > c.max_seq) return MB_ERR_SHAPE;
    if (!input_ids || !visual || !acoustic || !attention_mask || !token_type_ids) return MB_ERR_ARG;
    CK(xl_prepare_pass(e, B * L, st));
    PrologueArgs pa = {};
    e->fill_copies(pa, e->ws, input_ids, visual, acoustic, attention_mask, token_type_ids, labels, B, L, c.visual_dim, c.acoustic_dim,
                   c.num_labels);
    CK(step_prologue(pa, st));
    e->staged(e->ws, labels != nullptr, staged6);
    return MB_OK;
}

int mb_xlnet_graph_stats(const mb_xlnet_engine* e, size_t* captures, size_t* launches) {
    if (!e) return MB_ERR_ARG;
    if (captures) *captures = e->graph_captures;
    if (launches) *launches = e->graph_launches;
    return MB_OK;
}
size_t mb_xlnet_trainable_count(const mb_xlnet_engine* e) { return e->n_trainable; }

// ---------------------------------------------------------------------------------------------- query stream (target_mapping)
// xlnet.py:238-240, 306-313, 374-399: with target_mapping [B][M][L] XLNetModel runs a second stream g [M][B][H] next to h.  g starts as
// mask_emb, every layer projects it with the layer's q, maps the M query rows onto the L positions, attends over the CONTENT stream's
// keys / values / positions of that layer under attn_mask_g (data_mask without the i == j exemption), maps the result back to the M
// targets, then shares post_attention and the feed-forward block with h.  h never reads g, so the stream is a post-pass over what the
// last EVAL forward left in the workspace (every layer's q | k | v and kr); nothing of that pass is overwritten -- all intermediates
// live in the caller's scratch -- and MAG touches h only (xlnet.py:371-372).
struct XlQsLayout { size_t state, state_stride, qg, qkv, vec, vecg, s1, y1, u, gl, s2, st, xs, z, pooled, bytes; };
static XlQsLayout xl_qs_layout(const mb_xlnet_engine* e, int B, int M, int L) {
    const mb_xlnet_config& c = e->c;
    const size_t es = esize(c.dtype), H = c.d_model, I = c.d_inner;
    const size_t R = align_up((size_t)B * M, 128), T = align_up((size_t)B * L, 128);      // (row counts padded as the workspace's are)
    Carver w;
    XlQsLayout q;
    q.state_stride = align_up(R * H * es, 256);
    q.state = w.take(q.state_stride * (c.n_layer + 1));
    q.qg = w.take(R * H * es); q.qkv = w.take(T * 3 * H * es); q.vec = w.take(T * H * es); q.vecg = w.take(R * H * es);
    q.s1 = w.take(R * H * es); q.y1 = w.take(R * H * es); q.u = w.take(R * I * es); q.gl = w.take(R * I * es); q.s2 = w.take(R * H * es);
    q.st = w.take(R * 2 * 4);
    q.xs = w.take((size_t)B * H * es); q.z = w.take((size_t)B * H * 4); q.pooled = w.take((size_t)B * H * 4);
    q.bytes = w.off;
    return q;
}
size_t mb_xlnet_query_stream_scratch_bytes(const mb_xlnet_engine* e, int B, int M, int L) {
    if (!e || B < 1 || M < 1 || L < 1) return 0;
    return xl_qs_layout(e, B, M, L).bytes;
}
size_t mb_xlnet_query_stream_state_bytes(const mb_xlnet_engine* e, int B, int M) {
    if (!e || B < 1 || M < 1) return 0;
    return xl_qs_layout(e, B, M, 1).state_stride;
}
int mb_xlnet_query_stream(mb_xlnet_engine* e, const float* target_mapping, int M, void* scratch, size_t scratch_bytes, float* logits_g,
                          void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (!e || !target_mapping || !scratch) return MB_ERR_ARG;
    if (!e->P || !e->ws || !e->ran_forward) return MB_ERR_ARG;
    if (e->training || e->in_step || e->mems) return MB_ERR_MODE;          // eval passes without memories only
    const mb_xlnet_config& c = e->c;
    const int dt = c.dtype, H = c.d_model, I = c.d_inner, B = e->B, L = e->L, nh = c.n_head, NL = c.n_layer, R = B * M;
    if (M < 1 || M > c.max_seq) return MB_ERR_SHAPE;
    const XlQsLayout q = xl_qs_layout(e, B, M, L);
    if (scratch_bytes < q.bytes || ((uintptr_t)scratch & 255)) return MB_ERR_ARG;
    float* P = e->P;
    char* ws = e->ws;
    char* sc = (char*)scratch;
    auto state = [&](int i) { return sc + q.state + (size_t)i * q.state_stride; };
    float* mean = (float*)(sc + q.st);
    float* rstd = mean + align_up((size_t)R, 128);
    CK(xlnet_broadcast_row(dt, P + e->mask_emb, state(0), R, H, st));                                                    // xlnet.py:306-310
    for (int l = 0; l < NL; ++l) {
        const XlLayerOff& o = e->lo[l];
        const XlLayerWs& w = e->lw[l];
        const char* g = state(l);
        CK(gemm(dt, GEMM_NN, EPI_ADD_RES, R, H, H, g, H, e->W(o.q), H, sc + q.qg, H, nullptr, nullptr, nullptr, nullptr, 0, kNoDrop, 1, 0, st));
        CK(xlnet_map_query(dt, target_mapping, sc + q.qg, ws + w.qkv, sc + q.qkv, B, M, L, H, st));
        CK(xlnet_attention_forward(dt, sc + q.qkv, ws + w.kr, P + o.rwb, P + o.rrb, P + o.rsb, P + o.seg, e->seg, e->mask, sc + q.vec, nullptr,
                                   B, L, nh, kNoDrop, st, e->head_mask ? e->head_mask + (size_t)l * nh : nullptr, e->perm, 1));
        CK(xlnet_unmap_vec(dt, target_mapping, sc + q.vec, sc + q.vecg, B, M, L, H, st));
        // post_attention and the feed-forward block, the content stream's weights on M rows per sample
        CK(gemm(dt, GEMM_NT, EPI_BIAS_DROP_RES, R, H, H, sc + q.vecg, H, e->W(o.o), H, sc + q.s1, H, nullptr, nullptr, nullptr, g, H, kNoDrop, 1, 0, st));
        CK(ln_forward(dt, sc + q.s1, P + o.ralnw, P + o.ralnb, c.layer_norm_eps, sc + q.y1, mean, rstd, R, H, kNoDrop, st));
        CK(gemm(dt, GEMM_NT, EPI_BIAS_GELU, R, I, H, sc + q.y1, H, e->W(o.w1), H, sc + q.u, I, sc + q.gl, nullptr, P + o.b1, nullptr, 0, kNoDrop, 1, 0, st));
        CK(gemm(dt, GEMM_NT, EPI_BIAS_DROP_RES, R, H, I, sc + q.gl, I, e->W(o.w2), I, sc + q.s2, H, nullptr, nullptr, P + o.b2, sc + q.y1, H, kNoDrop, 1, 0, st));
        CK(ln_forward(dt, sc + q.s2, P + o.fflnw, P + o.fflnb, c.layer_norm_eps, state(l + 1), mean, rstd, R, H, kNoDrop, st));
    }
    if (!logits_g) return MB_OK;
    // the head on the query stream's output (xlnet.py:396-399 returns output_g first; 506-509: summary of its last row)
    CK(last_token_forward(dt, state(NL), sc + q.xs, B, M, H, kNoDrop, st));
    float* z = (float*)(sc + q.z);
    CK(gemm(dt, GEMM_NT, EPI_BIAS_F32, B, H, H, sc + q.xs, H, e->W(e->wsum), H, nullptr, H, nullptr, z, P + e->bsum, nullptr, 0, kNoDrop, 1, 64, st));
    CK(head_forward(z, P + e->wc, P + e->bc, nullptr, (float*)(sc + q.pooled), logits_g, nullptr, nullptr, B, H, c.num_labels, kNoDrop, st));
    return MB_OK;
}

const void* mb_xlnet_hidden_state(const mb_xlnet_engine* e, int i) {      // input of layer i (before the MAG injection), i = n_layer: last output
    if (!e || !e->ws || i < 0 || i > e->c.n_layer) return nullptr;
    return e->ws + e->ws_x[i];
}
const void* mb_xlnet_attention_probs(const mb_xlnet_engine* e, int layer, int* padded_len) {
    if (!e || !e->ws || !e->ran_forward || layer < 0 || layer >= e->c.n_layer) return nullptr;
    if (padded_len) *padded_len = e->L <= 32 ? 32 : (e->L <= 64 ? 64 : 128);
    return e->ws + e->lw[layer].psave;
}
const void* mb_xlnet_sequence_output(const mb_xlnet_engine* e) { return e->ws ? e->ws + e->ws_x[e->c.n_layer] : nullptr; }
int mb_xlnet_set_inputs_embeds(mb_xlnet_engine* e, const float* inputs_embeds) {
    if (!e) return MB_ERR_ARG;
    e->emb_in = inputs_embeds;
    return MB_OK;
}
const float* mb_xlnet_inputs_embeds_grad(const mb_xlnet_engine* e) {
    if (!e || !e->ws || !e->ran_forward) return nullptr;
    return (const float*)(e->ws + e->ws_demb);
}
// MAG_XLNetModel's return value (xlnet.py:396-405): the last layer's output AFTER the final dropout, whole sequence.  Written to a
// scratch activation (valid until the next backward); eval mode: a copy.
const void* mb_xlnet_model_output(mb_xlnet_engine* e, void* stream) {
    if (!e || !e->ws || !e->ran_forward) return nullptr;
    const mb_xlnet_config& c = e->c;
    if (drop_rows(c.dtype, e->ws + e->ws_x[c.n_layer], e->ws + e->ws_dxb, e->B * e->L, c.d_model, e->key(XS_FINAL, c.dropout), (hipStream_t)stream))
        return nullptr;
    return e->ws + e->ws_dxb;
}
// the autograd edge of the base model: d_output [B*L][H] (activation dtype) = gradient of mb_xlnet_model_output's tensor -> entry
// gradient of the layer stages (the final dropout's mask applied); then mb_xlnet_backward(stage 1 .. n_layer + 1)
int mb_xlnet_backward_outputs(mb_xlnet_engine* e, const void* d_output, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (!e || !e->G || !e->ran_forward || !d_output) return MB_ERR_ARG;
    CK(e->begin_backward_pass(e->G, st));
    const mb_xlnet_config& c = e->c;
    return drop_rows(c.dtype, d_output, e->ws + e->ws_dxa, e->B * e->L, c.d_model, e->key(XS_FINAL, c.dropout), st);
}

int mb_xlnet_stage_grad_ranges(const mb_xlnet_engine* e, int stage, size_t* offs, size_t* lens, int cap) {
    const int NL = e->c.n_layer;
    std::vector<std::pair<size_t, size_t>> r;
    auto span = [&](size_t a, size_t b) { r.push_back({a, b - a}); };
    if (stage == 0) span(e->wsum, e->sh_end);
    else if (stage <= NL) {
        const int l = NL - stage;
        auto wspan = [&](int k) { span(e->lo[k].q, k + 1 < NL ? e->lo[k + 1].q : e->wsum); };
        if (!e->deferred) wspan(l);
        else if (l + 1 < NL) wspan(l + 1);           // deferred join: a layer's weights are final one stage later
    } else if (stage == NL + 1) {
        if (e->deferred) span(e->lo[0].q, NL > 1 ? e->lo[1].q : e->wsum);
        span(e->small_decay_begin, e->n_decay);      // seg_embed / layer_norm weights, word embedding, MAG weights, logits_proj.weight
        span(e->n_decay, e->n_trainable);            // every no-decay parameter
    } else return -1;
    int n = 0;
    for (auto& p : r) { if (n < cap) { offs[n] = p.first; lens[n] = p.second; } ++n; }
    return n;
}

}  // extern "C"
