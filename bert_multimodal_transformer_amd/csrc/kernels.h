// Internal C++ launch API of the HIP kernels (host side).  The public C ABI is include/magbert_hip.h.
#pragma once
#include <cstdio>
#include <cstdlib>
#include "common.h"

namespace mb {

enum {
    MB_OK = 0,
    MB_ERR_SHAPE = 1001,     // unsupported shape / alignment
    MB_ERR_MODE = 1002,      // unsupported epilogue / layout combination
    MB_ERR_DTYPE = 1003,
    MB_ERR_ARG = 1004,
    MB_ERR_COMM = 1005,      // gradient exchange: RCCL missing / no backend (mb_comm_last_error() has the text); 2000 + n = ncclResult_t n
};

// MB_CK_TRACE=1: every failing call of a CK(...) chain is printed with its source line (a bare HIP error code from the middle of a
// 200-kernel step says nothing about which call produced it)
inline void ck_trace(const char* expr, const char* file, int line, int code) {
    static int on = -1;
    if (on < 0) { const char* v = getenv("MB_CK_TRACE"); on = (v && atoi(v)) ? 1 : 0; }
    if (on) fprintf(stderr, "[magbert] %s:%d: %s -> %d\n", file, line, expr, code);
}

// ------------------------------------------------------------------------------------------ GEMM
enum { GEMM_NT = 0,   // A [M][K] row, B [N][K] row      : Y = X W^T            (forward Linear)
       GEMM_NN = 1,   // A [M][K] row, B [K][N] kmaj     : dX = dY W            (dgrad)
       GEMM_TN = 2 }; // A [K][M] kmaj, B [K][N] kmaj    : dW = dY^T X          (wgrad)

enum { EPI_BIAS = 0,          // C = alpha*acc + bias                                   (T out)
       EPI_BIAS_GELU = 1,     // u = acc + bias ; C = gelu'(u) ; C2 = gelu(u) [* dropout]  (T out x2)
       EPI_BIAS_DROP_RES = 2, // C = dropout(acc + bias) + R                            (T out)
       EPI_ADD_RES = 3,       // C = acc (+ R)                                          (T out)
       EPI_DGELU = 4,         // C = acc * R [* dropout], R = gelu'(u) saved by mode 1  (T out)
       EPI_ACCUM_F32 = 5,     // Cf += acc   (atomic when split-K)                      (fp32 out)
       EPI_BIAS_F32 = 6,      // Cf = alpha*acc + bias                                  (fp32 out)
       // EXPERIMENT (MB_ADAMW_IN_WGRAD=1, grouped weight gradients of a single-process step whose gradient buffer is known-zero): the tile
       // is not stored as a gradient at all -- the epilogue applies HF-AdamW (adamw.hip's arithmetic) to its own [BM][BN] patch of
       // the parameters: C = p, C2 = m, R = v (fp32, the gradient's layout), colsum = bf16 shadow (or null), bias = AdamArgs in
       // device memory.  Saves the gradient's round trip through HBM (8 of 30 B/parameter); measured in profiles/r05_adamw_in_wgrad_ab.txt
       EPI_WGRAD_ADAM = 7 };

struct AdamArgs {
    float lr, beta1, beta2, eps, weight_decay, step_size;   // step_size = lr*sqrt(1-b2^t)/(1-b1^t)
    float grad_scale;                                        // multiply g before use (1/world_size for DP averaging)
};

struct GemmArgs {
    const void* A; const void* B;
    int M, N, K, lda, ldb;
    void* C; int ldc;          // T output (modes 0-4)
    void* C2;                  // second T output (mode 1), same ldc
    float* Cf;                 // fp32 output (modes 5, 6), ldc
    const float* bias;         // [N] fp32 or null
    float* colsum;             // EPI_DGELU: if non-null, colsum[n] += sum_m C[m][n] (bias grad of the previous Linear)
    const void* R; int ldr;    // residual / aux input (T)
    float alpha;
    DropKey drop;
    int kchunk;                // filled by the launcher
    int dbg;                   // ablation switches (MB_GEMM_DBG): 1 = no DMA issue, 2 = no MFMA, 4 = no LDS fragment reads
    int reg_m, reg_n, tpr_m, tpr_n;   // XCD regions (filled by the launcher): reg_m*reg_n == 8, tiles per region
    int overwrite;                    // EPI_ACCUM_F32 without split-K: Cf = acc instead of Cf += acc (the caller knows Cf holds zeros)
    int cvalid;                       // EPI_ACCUM_F32, > 0: only columns [0, cvalid) of the N (padded) ones are stored, one dword per lane -- Cf / ldc
                                      // need no alignment then (MAG's weight gradients go straight into tensors 815 / 842 / 47 / 74 floats wide)
    unsigned long long* trace;        // MB_GEMM_TRACE=1: [blocks][8] wall-clock stamps (100 MHz) of the phases of every block, else null
    // Segmented B (0 = off): B's contiguous dimension (the columns n of a k-major B, the k of a row-major B) is cut into pieces of
    // `bseg` elements that live in separate tensors `bseg_stride` elements apart: element (r, c) sits at
    // B + (c / bseg) * bseg_stride + r * ldb + c % bseg.  This is how MAG-XLNet's q | k | v projections -- three [768][768] tensors
    // next to each other in the flat parameter buffer -- run as ONE forward GEMM (N = 2304) and ONE dgrad (K = 2304) instead of
    // three each.  bseg must be a multiple of the tile (k-major) / of the k-stage (row-major); no split-K.
    int bseg; size_t bseg_stride;
    GradAcc acc;                      // where colsum goes in deterministic mode (common.h); {} = fp32 atomics
};
// copies the stamps of the last traced launch to the host (measurement tooling: tools/gemm_bench --trace); returns the block count
int gemm_trace_fetch(unsigned long long* host_out, int max_blocks);

// tile: 0 = auto, 64 or 128.  splits: split-K factor (only EPI_ACCUM_F32).
int gemm_launch(int dtype, int layout, int mode, const GemmArgs& a, int splits, int tile, hipStream_t st);

// Grouped wgrad (GEMM_TN, EPI_ACCUM_F32): `count` <= MB_MAX_GROUP (8) problems Cf_g[M_g][N_g] += A_g^T B_g in one launch.
// Needs whole tiles and whole 128-byte K rows for every problem (gemm_grouped_tn_ok tells); tile = 64 | 128.
#define MB_MAX_GROUP 8
// An HF-AdamW update riding inside a grouped weight-gradient launch (MB_ADAMW_RIDE): `blocks` extra workgroups at the head of the grid
// (a multiple of 8: the tiles' XCD placement is unchanged) update n4 quads of weight-decayed parameters -- the GEMM weights of the
// layer whose gradients the PREVIOUS launch completed -- while the other workgroups multiply.  The update is HBM-bound, the tiles
// are not, and the 256 x 128 launch leaves 40 of the 256 CUs without a tile.  p / g / m / v / shadow point at the range's first element.
struct AdamRide {
    float *p, *g, *m, *v;
    bf16* shadow;             // bf16 operand shadow of the range or null
    size_t n4;
    const AdamArgs* dyn;      // this step's scalars (device memory: the step prologue writes them)
    int blocks;               // 0 = no rider
    int zero_grad;            // 0: the next backward overwrites these gradients (engine_common.h keep_in_step)
};
void gemm_log_ride(const AdamRide& r);      // MB_GEMM_LOG=1: "[magbert ride] params=... blocks=..." for a launch that carries riders (gemm.hip)
struct GroupedGemmArgs {
    GemmArgs g[MB_MAX_GROUP];
    int first[MB_MAX_GROUP + 1];    // first block (region placement) or first tile of the group-wide list (chunk > 0) of every problem
    int count;
    int chunk;                      // > 0: tiles per XCD of the group-wide XCD-compact placement (gemm.hip); 0: per-problem regions
    AdamRide ride;                  // ride.blocks > 0: that many workgroups in front of the tiles run an optimizer update instead
};
// A dgrad launch (GEMM_NN; mode EPI_ADD_RES or EPI_DGELU) with riders: gemm_nn_ride_tiles -> the tile count (padded to 8) of the launch
// gemm_launch would make of `a` when that is one of the two kernels that leave block slots free (64 x 64 three-slot, 128 x 128 two-slot;
// *per_cu = block slots per CU), else 0; gemm_nn_ride_launch: that launch with ride.blocks (a multiple of 8) rider workgroups behind the
// tiles (MB_ERR_MODE when gemm_nn_ride_tiles says 0).
int gemm_nn_ride_tiles(int dtype, int mode, const GemmArgs& a, int* per_cu);
int gemm_nn_ride_launch(int dtype, int mode, const GemmArgs& a, const AdamRide& ride, hipStream_t st);
int gemm_grouped_tn_ok(int dtype, const GemmArgs* probs, int count, int tile);
int gemm_grouped_tn_launch(int dtype, const GemmArgs* probs, int count, int tile, hipStream_t st, int stages = 0, bool adam = false,
                           const AdamRide* ride = nullptr);   // stages: 0 = MB_GROUP_STAGES, 4 | 5 = deeper ring (64 x 64 tiles only)

// ------------------------------------------------------------------------------------------ row kernels (rowops.hip)
// LayerNorm over the last dim H (H % 256 == 0, H <= 1024): y = (x-mean)*rstd*gamma + beta ; optional dropout on y.
int ln_forward(int dtype, const void* x, const float* gamma, const float* beta, float eps, void* y,
               float* mean, float* rstd, int rows, int H, DropKey drop, hipStream_t st,
               Prefetch pf = {nullptr, 0, nullptr});
// LayerNorm backward. dy: grad of LN output (after optional dropout `drop_out`), x: saved LN input.
//   dx      = grad wrt LN input                         (T)  [required]
//   dx_drop = dx * mask(drop_in)                        (T)  [optional: grad of the pre-dropout Linear output]
//   dgamma, dbeta += column sums ; dbias += colsum(dx_drop or dx)   (fp32, atomics) [each optional]
int ln_backward(int dtype, const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                void* dx, void* dx_drop, float* dgamma, float* dbeta, float* dbias,
                int rows, int H, DropKey drop_out, DropKey drop_in, hipStream_t st);
// folds a deterministic-mode shadow accumulator into the fp32 gradients g[begin, end) and clears it (common.h GradAcc)
int grad_fold(GradAcc acc, float* g, size_t begin, size_t end, hipStream_t st);
// same, but the three column sums go to per-block partial slabs partials[nblk][3][H] (no atomics); returns nblk through
// *nblk.  ln_reduce_partials adds them into up to 6 destinations in one launch (two LayerNorms of a layer).
size_t ln_partials_floats(int rows, int H);
int ln_backward_partials(int dtype, const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                         void* dx, void* dx_drop, float* partials, int* nblk, int rows, int H, DropKey drop_in, hipStream_t st,
                         Prefetch pf = {nullptr, 0, nullptr});
int ln_reduce_partials(const float* partials_a, const float* partials_b, int nblk, int H, float* const* dst6, hipStream_t st, GradAcc acc = {});
// The same for `layers` layers in ONE launch: layer l's slabs start at partials_x + l * layer_stride floats, its six destinations
// are dst[l][0..5].  (A single-process step has no use for a layer's LayerNorm / bias gradients before AdamW: twelve 5-us
// launches become one.)
#define MB_LN_MAX_LAYERS 32
struct LnReduceDst { float* d[MB_LN_MAX_LAYERS][6]; int nblk[MB_LN_MAX_LAYERS]; };    // nblk[slot] > 0: that slot's own slab count
int ln_reduce_partials_layers(const float* partials_a, const float* partials_b, size_t layer_stride, int layers, int nblk, int H,
                              const LnReduceDst& dst, hipStream_t st, GradAcc acc = {});

// BertEmbeddings: e = dropout(LN(word[ids] + pos[l] + type[seg])).
int embed_ln_forward(int dtype, const int64_t* ids, const int64_t* seg, const float* word, const float* pos,
                     const float* type, const float* gamma, const float* beta, float eps, void* out,
                     float* mean, float* rstd, int B, int L, int H, DropKey drop, hipStream_t st,
                     const int64_t* pos_ids = nullptr);       // [B*L] rows of `pos` (null: arange(L) per sample, bert.py:211-216)
int embed_ln_backward(int dtype, const void* dout, const int64_t* ids, const int64_t* seg, const float* word,
                      const float* pos, const float* type, const float* gamma, const float* mean, const float* rstd,
                      float* dsum_ws, float* dword, float* dpos, float* dtype_, float* dgamma, float* dbeta,
                      int B, int L, int H, int pad_id, DropKey drop, hipStream_t st, const int64_t* pos_ids = nullptr, GradAcc acc = {},
                      // part: dgamma / dbeta go to rows 0 / 1 of a partial set [nblk][3][H] instead (reduced by ln_reduce_partials*)
                      float* part = nullptr, int* nblk = nullptr,
                      // part_b (with part, default positions only): ONE launch that also produces dpos / dtype; dtype[0] -> row 2 of `part`,
                      // dtype[1] -> row 0 of `part_b` (the caller's reduction adds them to dtype_), *nblk = L * ceil(B / 8)
                      float* part_b = nullptr,
                      // id_count: the table the step prologue of THIS batch counted token ids into (vocab_size ints, zero between steps):
                      // rows whose id occurs once add their word-embedding gradient without atomics; the entries are cleared again
                      int* id_count = nullptr);

// column sums: out[n] += sum_m x[m][n]
int colsum(int dtype, const void* x, int ldx, float* out, int rows, int cols, hipStream_t st, GradAcc acc = {});

// fp32 [rows][cols] -> T [rows][cols_pad] zero padded (modality tensors -> MFMA operands)
int pack_pad(int dtype, const float* src, int cols, void* dst, int cols_pad, int rows, hipStream_t st);
// fp32 -> T contiguous conversion (n elements)
int convert(int dtype, const float* src, void* dst, size_t n, hipStream_t st);
int widen(int dtype, const void* src, float* dst, size_t n, hipStream_t st);       // dst fp32 <- src (dtype), n % 4 == 0

// ------------------------------------------------------------------------------------------ MAG (mag.hip)
struct MagDims { int T, H, V, A, Vp, Ap; };
// pack master weights (reference layout) into the MFMA operand layout:
//   We [2H][H]  = [W_hv[:, V:] ; W_ha[:, A:]]     Wv [2H][Vp] = [W_hv[:, :V] ; W_v]     Wa [2H][Ap] = [W_ha[:, :A] ; W_a]
int mag_pack_weights(int dtype, const float* W_hv, const float* W_ha, const float* W_v, const float* W_a,
                     void* We, void* Wv, void* Wa, MagDims d, hipStream_t st);
// scatter-add packed weight grads back into the reference layout
int mag_unpack_wgrads(const float* dWe, const float* dWv, const float* dWa, float* dW_hv, float* dW_ha,
                      float* dW_v, float* dW_a, MagDims d, hipStream_t st);
// gate + norm-ratio clamp + residual + LayerNorm + dropout (modeling.py:27-49) from the three pre-activation panels
int mag_gate_forward(int dtype, const void* e, const void* Ze, const void* Zv, const void* Za,
                     const float* b_hv, const float* b_ha, const float* b_v, const float* b_a,
                     const float* gamma, const float* beta, float ln_eps, float beta_shift,
                     void* out, float* mean, float* rstd, MagDims d, DropKey drop, hipStream_t st);
int mag_gate_backward(int dtype, const void* dout, const void* e, const void* Ze, const void* Zv, const void* Za,
                      const float* b_hv, const float* b_ha, const float* b_v, const float* b_a,
                      const float* gamma, const float* mean, const float* rstd, float beta_shift,
                      void* de, void* dZe, void* dZv, void* dZa,
                      float* db_hv, float* db_ha, float* db_v, float* db_a, float* dgamma, float* dbeta,
                      MagDims d, DropKey drop, hipStream_t st, GradAcc acc = {},
                      // part_a / part_b: instead of adding the six column sums to db_* / dgamma / dbeta, every block writes its slab
                      // of two partial sets [nblk][3][H] ({db_hv, db_ha, db_v}, {db_a, dgamma, dbeta}) for ln_reduce_partials(_layers)
                      float* part_a = nullptr, float* part_b = nullptr, int* nblk = nullptr);

// ------------------------------------------------------------------------------------------ attention (attention.hip)
// qkv: [B*L][3H] token-major (q | k | v, head h at columns h*64..), mask: int64 [B][L] (1 = attend),
// ctx: [B*L][H].  softmax(QK^T/sqrt(dh) + (1-mask)*-10000) -> dropout -> . V   (dh = 64, L <= 128)
// probs (fp32 [B][nh][L][L], may be null): the attention probabilities after dropout and head mask (output_attentions)
// head_scale (fp32 [nh], may be null): head_mask of this layer -- the dropped probabilities of head h are multiplied by it
int attention_forward(int dtype, const void* qkv, const int64_t* mask, void* ctx, int B, int L, int nh,
                      DropKey drop, hipStream_t st, float* probs = nullptr, const float* head_scale = nullptr);
// dbias (fp32 [3H], may be null): += column sums of dqkv (bias grads of the fused QKV Linear)
int attention_backward(int dtype, const void* qkv, const int64_t* mask, const void* ctx, const void* dctx,
                       void* dqkv, float* dbias, int B, int L, int nh, DropKey drop, hipStream_t st,
                       const float* head_scale = nullptr, GradAcc acc = {},
                       const struct AdamRide* ride = nullptr);      // bf16: AdamW riders behind the (batch, head) workgroups (AdamRide below)
int attention_backward_free_slots(int dtype, int L, int nblk, int cus);      // workgroups that fit the last round of that launch
int attention_trace_fetch(unsigned long long* host_out, int max_blocks);     // MB_ATTN_TRACE=1: stamps of the last attention_backward

// ------------------------------------------------------------------------------------------ XLNet (xlnet_attention.hip, xlnet_rowops.hip)
// relative attention core, L <= 128.  qkv [T][3H] token-major, kr [B][2L][H], psave/gsave [B][nh][LP][LP] (LP = 32 | 64 | 128 >= L).
int xlnet_attention_forward(int dtype, const void* qkv, const void* kr, const float* r_w_bias, const float* r_r_bias,
                            const float* r_s_bias, const float* seg_embed, const int64_t* seg, const int64_t* mask, void* vec,
                            void* psave, int B, int L, int nh, DropKey drop, hipStream_t st, const float* head_scale = nullptr,
                            const uint8_t* perm = nullptr,       // perm [B][L][L] bytes or null: != 0 <=> query i may not attend to key j (xlnet.py:265-296)
                            int gstream = 0);                    // 1: the query stream's mask (attn_mask_g = data_mask, no i == j exemption; xlnet.py:288-296)
int xlnet_attention_backward(int dtype, const void* qkv, const void* kr, const float* r_w_bias, const float* r_r_bias,
                             const float* r_s_bias, const float* seg_embed, const int64_t* seg, const int64_t* mask,
                             const void* psave, const void* dvec, void* gsave, void* dqkv, void* dkr, float* d_rwb,
                             float* d_rrb, float* d_rsb, float* d_seg, int B, int L, int nh, DropKey drop, hipStream_t st,
                             const float* head_scale = nullptr,      // head_scale [nh] or null: head_mask of the layer
                             GradAcc acc = {},                       // deterministic mode: the four bias / seg_embed column sums (common.h)
                             const struct AdamRide* ride_q = nullptr, const struct AdamRide* ride_kv = nullptr);      // bf16, L <= 64: riders of the two launches
int xlnet_attention_backward_free_slots(int dtype, int L, int nblk, int cus);
// out[t] = dropout(word[ids[t]])  (xlnet.py:304-305) ; backward scatter-adds into dword
int gather_drop_forward(int dtype, const int64_t* ids, const float* word, void* out, int rows, int H, DropKey drop, hipStream_t st);
int gather_drop_backward(int dtype, const void* dout, const int64_t* ids, float* dword, int rows, int H, DropKey drop, hipStream_t st, GradAcc acc = {});
// (both: ids == nullptr = inputs_embeds -- `word` / `dword` are then [rows][H] fp32, read / written row by row)
// dst[i] += src[i], fp32, n % 4 == 0
int add_f32(float* dst, const float* src, size_t n, hipStream_t st);
// y = x * dropout mask over [rows][H] (element index = offset)
int drop_rows(int dtype, const void* x, void* y, int rows, int H, DropKey drop, hipStream_t st);
// pos[b][p][:] = dropout([sin(pos_p * inv_freq) | cos(...)]) with pos_p = L - p, p in [0, 2L)   (xlnet.py:93-146,332-333)
int xlnet_pos_emb(int dtype, void* out, int B, int L, int H, DropKey drop, hipStream_t st);
// xs[b] = x[b, L-1, :] * dropout   (final dropout xlnet.py:396 + SequenceSummary "last") ; backward scatters into a zeroed dx
int last_token_forward(int dtype, const void* x, void* xs, int B, int L, int H, DropKey drop, hipStream_t st);
int last_token_backward(int dtype, const void* dxs, void* dx, int B, int L, int H, DropKey drop, hipStream_t st);
// query stream under target_mapping tm [B][M][L] fp32 (xlnet.py:306-313, 374-399): the mask_emb row broadcast to [rows][H]; the
// attention operand of the g stream (q = tm-mapped qg [B*M][H] onto the L positions, k | v copied from the content stream's
// [T][3H]); the attention output mapped back to the M targets
int xlnet_broadcast_row(int dtype, const float* row, void* out, int rows, int H, hipStream_t st);
int xlnet_map_query(int dtype, const float* tm, const void* qg, const void* qkv_h, void* qkv_g, int B, int M, int L, int H, hipStream_t st);
int xlnet_unmap_vec(int dtype, const float* tm, const void* vec, void* vecg, int B, int M, int L, int H, hipStream_t st);

// ------------------------------------------------------------------------------------------ head (head.hip)
// pooled = tanh(z) ; logits = dropout(pooled) Wc^T + bc ; optional MSE loss (mean over B*nl) accumulated into loss[0]
// and (if non-null) into the running sum loss_run[0].
int head_forward(const float* z, const float* Wc, const float* bc, const float* labels, float* pooled,
                 float* logits, float* loss, float* loss_run, int B, int H, int nl, DropKey drop, hipStream_t st);
// dlogits (given, or MSE grad if labels != null: 2*(logit-y)/(B*nl)*loss_scale) -> dz (grad wrt pooler pre-activation),
// dWc, dbc accumulated.  dz is written in the activation dtype (operand of the pooler dgrad / wgrad GEMMs).
int head_backward(int dtype, const float* dlogits, const float* logits, const float* labels, float loss_scale,
                  const float* pooled, const float* Wc, void* dz, float* dWc, float* dbc,
                  int B, int H, int nl, DropKey drop, hipStream_t st, GradAcc acc = {}, float* dbp = nullptr,
                  void* zero_p = nullptr, size_t zero_bytes = 0);   // dbp += column sums of dz; zero_p[0, zero_bytes) cleared by extra blocks

// ------------------------------------------------------------------------------------------ optimizer (adamw.hip)
// transformers 3.0.2 AdamW over flat fp32 buffers; elements [0, n_decay) use weight_decay, the rest 0.
// shadow (bf16, may be null): shadow[i] = bf16(p[i]) for i in [sh_begin, sh_end). zero_grad: g <- 0 after use.
// dyn (device pointer, may be null): when set, the hyper-parameters are read from *dyn instead of `a` (replayed step graphs)
// keep range [keep_begin, keep_end) (elements, multiples of 4; empty by default): g is NOT zeroed there -- the caller knows the next
// backward overwrites that range (the layers' GEMM weight gradients of a single-call step: 340 MB of zeros per step nobody reads)
int adamw_step(float* p, float* g, float* m, float* v, void* shadow, size_t n, size_t n_decay,
               size_t sh_begin, size_t sh_end, AdamArgs a, int zero_grad, hipStream_t st, const AdamArgs* dyn = nullptr,
               size_t keep_begin = 0, size_t keep_end = 0);

// ------------------------------------------------------------------------------------------ step prologue (rowops.hip)
// Everything that changes from one optimizer step to the next, moved into device memory by ONE small launch so that the rest
// of the step can be a replayed hipGraph: the six batch tensors (gathered into the engine's fixed staging buffers), the
// dropout keys of every site for (seed, step) -- same derivation as make_key() on the host -- and the AdamW scalars of the
// two parameter groups (lr, bias-corrected step size, gradient scale).
#define MB_PROLOGUE_MAX_COPIES 8
struct PrologueArgs {
    const uint32_t* src[MB_PROLOGUE_MAX_COPIES]; uint32_t* dst[MB_PROLOGUE_MAX_COPIES]; uint32_t dwords[MB_PROLOGUE_MAX_COPIES];
    int ncopies;
    uint64_t seed, step;
    uint32_t* keys; int nsites;            // keys[2 * site + {0, 1}]
    AdamArgs adam[2]; AdamArgs* adam_dst;  // may be null
    uint32_t* zero_dw;                     // one dword cleared by the launch (the step's loss accumulator), may be null
    // modality tensors packed on the way in: src fp32 [rows][cols] (device or pinned host) -> dst [rows][pitch] of `dtype` (MAG's
    // GEMM operands; columns [cols, pitch) are never written and stay zero) -- what pack_pad does, without its two launches
    struct PackJob { const float* src; void* dst; int rows, cols, pitch, dtype; } pack[2];
    int npack;
    const int64_t* ids; int n_ids; int* id_count;     // id_count[ids[i]] += 1 (see embed_ln_backward), may be null
    // MAG's weights packed into the operands of its regrouped GEMMs (mag_pack.h) by EXTRA blocks of this launch, which is waiting for
    // PCIe anyway: the first `copy_blocks` blocks (filled by step_prologue) do everything above, the rest this.  W_hv == null: none
    struct MagPackW { const float* W_hv; const float* W_ha; const float* W_v; const float* W_a; void* We; void* Wv; void* Wa; MagDims d; int dtype; } magw;
    int copy_blocks;
};
int step_prologue(const PrologueArgs& a, hipStream_t st);
// p[0, bytes) = 0 as a kernel launch (bytes and p multiples of 4); up to MB_ZERO_MAX ranges in one launch
#define MB_ZERO_MAX 8
struct ZeroRanges {
    uint32_t* p[MB_ZERO_MAX]; size_t ndw[MB_ZERO_MAX]; int n;
    void add(void* ptr, size_t bytes) { if (bytes && n < MB_ZERO_MAX) { p[n] = (uint32_t*)ptr; ndw[n] = bytes / 4; ++n; } }
    bool full() const { return n >= MB_ZERO_MAX; }      // callers that collect an unbounded number of ranges flush (zero_fill_ranges) and reset at this point
};
int zero_fill(void* p, size_t bytes, hipStream_t st);
int zero_fill_ranges(const ZeroRanges& z, hipStream_t st);

}  // namespace mb
