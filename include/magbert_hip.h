/* magbert_hip.h -- C ABI of libmagbert_hip.so: the MI355X (gfx950) MAG-BERT training hot path.
 *
 * Drop-in boundary.  The reference (WasifurRahman/BERT_multimodal_transformer) is pure Python: its "FFI" for this
 * path is torch.nn.Module.forward / autograd / optimizer.step.  This library is what sits under those surfaces
 * (INTEGRATION.md shows the ctypes binding a maintainer adds to bert.py / modeling.py / multimodal_driver.py).
 * Every entry point is extern "C", takes plain device pointers + sizes and a hipStream_t (as void*), allocates
 * nothing on the device, never synchronises, and returns 0 on success or a non-zero code (mb_error_string()).
 * Device memory, streams and torch.distributed stay with the caller (PyTorch-ROCm is plumbing only).
 *
 * dtype: 0 = fp32 "parity mode" (exact-fp32 MFMA, logits within 1e-3 of the CPU reference),
 *        1 = bf16 "perf mode" (bf16 activations / MFMA operands, fp32 accumulate, fp32 master weights).
 * All row-major.  T = B*L tokens, H = 768, heads of 64.  Reference citations are relative to /root/reference.
 */
#ifndef MAGBERT_HIP_H
#define MAGBERT_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MB_DT_F32 0
#define MB_DT_BF16 1

/* dropout site key: mask(idx) = hash32(idx, k0, k1) < thresh ? drop : keep*scale.  thresh = 0 disables. */
typedef struct { uint32_t k0, k1, thresh; float scale; } mb_dropkey;

const char* mb_error_string(int code);
int mb_version(void);
/* fills the key for (seed, step, site) and probability p; p = 0 -> disabled */
void mb_make_dropkey(uint64_t seed, uint64_t step, uint32_t site, float p, mb_dropkey* out);

/* ------------------------------------------------------------------------------------------------ operators */

/* C[M,N] = sum_k A(m,k) B(n,k) with fused epilogue.  Replaces torch.nn.Linear forward/backward (addmm / mm)
 * under BertSelfAttention / BertSelfOutput / BertIntermediate / BertOutput (transformers 3.0.2, reached from
 * bert.py:221-229) and MAG's Linears (modeling.py:15-19).
 * layout 0 (NT): A[M][K], B[N][K]          forward  Y = X W^T
 * layout 1 (NN): A[M][K], B[K][N]          dgrad    dX = dY W
 * layout 2 (TN): A[K][M], B[K][N]          wgrad    dW += dY^T X   (fp32 accumulate, split-K)
 * epilogue: 0 C=alpha*acc+bias | 1 u=acc+bias: C=gelu'(u), C2=gelu(u) | 2 C=dropout(acc+bias)+R | 3 C=acc+R | 4 C=acc*R (R = the gelu'(u) saved by 1)
 *           5 Cf+=acc (fp32) | 6 Cf=alpha*acc+bias (fp32)
 * With epilogue 4, Cf (fp32 [N], may be NULL) receives += the column sums of C: the bias gradient of the Linear in front.
 * tile: 0 = chosen by shape (what the engines pass) | 64 | 128 | 12864 (128 x 64, four waves) | 256 (256 x 128 eight-wave ping-pong, bf16)
 *       | 12872 (128 x 64 eight-wave ping-pong, bf16: the choice for N = 768 at one tile per CU) | 25672 (256 x 64, the same at T = 4096);
 *       a tile that cannot take the problem
 *       (dtype, epilogue, k range) falls back to the next smaller one.  splits > 1: split-K (layout 2 only). */
int mb_gemm(int dtype, int layout, int epilogue, int M, int N, int K, const void* A, int lda, const void* B, int ldb,
            void* C, int ldc, void* C2, float* Cf, const float* bias, const void* R, int ldr, float alpha,
            const mb_dropkey* drop, int splits, int tile, void* stream);

/* Measurement hook (tools/gemm_bench --trace; not used by the product path): with MB_GEMM_TRACE=1 in the environment every
 * block of an mb_gemm launch stamps the 100 MHz wall clock at 0 entry, 1 first operand stage landed, 2 k loop done,
 * 3 epilogue issued, 4 its stores completed.  Copies the [blocks][8] stamps of the LAST launch to host_out (blocks that
 * exited without a tile stay 0) and returns the block count (0 when tracing is off). */
int mb_debug_gemm_trace(unsigned long long* host_out, int max_blocks);
/* same for mb_attention_backward with MB_ATTN_TRACE=1: 0 entry, 1 operands staged, 2 query sweep done, 3 dQ bias flushed,
 * 4 key sweep done, 5 exit */
int mb_debug_attention_trace(unsigned long long* host_out, int max_blocks);

/* `count` (<= 4) weight gradients dW_g[M_g][N_g] += dY_g[K][M_g]^T X_g[K][N_g] in ONE launch (fp32 accumulate) -- the
 * four torch.nn.Linear weight gradients autograd produces per BertLayer (mm_backward under loss.backward(),
 * multimodal_driver.py:378).  Every M_g, N_g must be a multiple of `tile` (64 | 128 | 256 = the 256 x 128 ping-pong tile, bf16: M_g % 256, N_g % 128) and K a
 * multiple of 128 bytes. */
int mb_gemm_grouped_wgrad(int dtype, int count, const int* M, const int* N, int K, const void* const* dY, const int* ldy,
                          const void* const* X, const int* ldx, float* const* dW, const int* ldw, int tile, void* stream);

/* dst[i] = (dtype) src[i] and back, n % 4 == 0 -- the gradient wire format of the data-parallel exchange in bf16 perf mode
 * (distributed.GradReducer: fp32 flat gradients -> bf16 staging -> RCCL all-reduce -> fp32).  New relative to the reference. */
int mb_narrow(int dtype, const float* src, void* dst, size_t n, void* stream);
int mb_widen(int dtype, const void* src, float* dst, size_t n, void* stream);

/* LayerNorm (+ dropout on the output) forward / backward -- torch.nn.LayerNorm under BertSelfOutput/BertOutput. */
int mb_layernorm_forward(int dtype, const void* x, const float* gamma, const float* beta, float eps, void* y,
                         float* mean, float* rstd, int rows, int H, const mb_dropkey* drop, void* stream);
int mb_layernorm_backward(int dtype, const void* dy, const void* x, const float* gamma, const float* mean,
                          const float* rstd, void* dx, void* dx_drop, float* dgamma, float* dbeta, float* dbias,
                          int rows, int H, const mb_dropkey* drop_out, const mb_dropkey* drop_in, void* stream);

/* BertEmbeddings (bert.py:81,211-216): LN(word[ids] + pos[0..L) + type[seg]) -> dropout. ids/seg int64 [B][L]. */
int mb_embed_forward(int dtype, const int64_t* ids, const int64_t* seg, const float* word, const float* pos,
                     const float* type, const float* gamma, const float* beta, float eps, void* out, float* mean,
                     float* rstd, int B, int L, int H, const mb_dropkey* drop, void* stream);
int mb_embed_backward(int dtype, const void* dout, const int64_t* ids, const int64_t* seg, const float* word,
                      const float* pos, const float* type, const float* gamma, const float* mean, const float* rstd,
                      float* dsum_ws, float* dword, float* dpos, float* dtype_, float* dgamma, float* dbeta, int B, int L,
                      int H, int pad_id, const mb_dropkey* drop, void* stream);

/* BertSelfAttention core (bert.py:221-229): ctx = dropout(softmax(QK^T/8 + (1-mask)*-1e4)) V.
 * qkv [T][3H] token-major, mask int64 [B][L], ctx/dctx [T][H], dqkv [T][3H].  L <= 128, head dim 64. */
int mb_attention_forward(int dtype, const void* qkv, const int64_t* mask, void* ctx, int B, int L, int nh,
                         const mb_dropkey* drop, void* stream);
int mb_attention_backward(int dtype, const void* qkv, const int64_t* mask, const void* dctx, void* dqkv, int B, int L,
                          int nh, const mb_dropkey* drop, void* stream);

/* Multimodal Adaptation Gate, MAG.forward (modeling.py:25-51) and its adjoint, for T tokens.
 * Parameters in the REFERENCE layout: W_hv [H][V+H], W_ha [H][A+H], W_v [H][V], W_a [H][A], biases [H], LayerNorm [H].
 * text [T][H] in `dtype`; visual [T][V], acoustic [T][A] fp32 (as the DataLoader yields them).
 * ws: caller scratch of mb_mag_workspace_bytes(); it also carries the activations saved for the backward. */
size_t mb_mag_workspace_bytes(int dtype, int T, int H, int V, int A);
int mb_mag_forward(int dtype, const void* text, const float* visual, const float* acoustic, const float* W_hv,
                   const float* b_hv, const float* W_ha, const float* b_ha, const float* W_v, const float* b_v,
                   const float* W_a, const float* b_a, const float* ln_w, const float* ln_b, float beta_shift,
                   const mb_dropkey* drop, void* out, void* ws, int T, int H, int V, int A, void* stream);
/* grads are ACCUMULATED (+=) into dW_*, db_*, dln_*; d_text [T][H] (dtype) is written; d_visual / d_acoustic
 * (fp32 [T][V] / [T][A], may be NULL) are written. */
int mb_mag_backward(int dtype, const void* d_out, const void* text, const float* W_hv, const float* b_hv,
                    const float* W_ha, const float* b_ha, const float* W_v, const float* b_v, const float* W_a,
                    const float* b_a, const float* ln_w, float beta_shift, const mb_dropkey* drop, void* ws,
                    void* d_text, float* d_visual, float* d_acoustic, float* dW_hv, float* db_hv, float* dW_ha,
                    float* db_ha, float* dW_v, float* db_v, float* dW_a, float* db_a, float* dln_w, float* dln_b,
                    int T, int H, int V, int A, void* stream);

/* transformers 3.0.2 AdamW.step over flat fp32 buffers (multimodal_driver.py:345,384). [0,n_decay) decays.
 * shadow: optional bf16 copy of p written for [sh_begin, sh_end).  zero_grad != 0 clears g (optimizer.zero_grad). */
int mb_adamw_step(float* p, float* g, float* m, float* v, void* shadow, size_t n, size_t n_decay, size_t sh_begin,
                  size_t sh_end, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                  int correct_bias, float grad_scale, int zero_grad, void* stream);

/* ------------------------------------------------------------------------------------------------ MAG-BERT engine
 * The whole MAG_BertForSequenceClassification forward / backward (bert.py:240-324 -> :76-237 -> modeling.py) as a
 * native step executor: one call enqueues every kernel of the pass on the stream (no Python between launches).   */
typedef struct {
    int vocab_size, hidden_size, num_layers, num_heads, intermediate_size, max_position, type_vocab, num_labels;
    int visual_dim, acoustic_dim, pad_token_id;
    float layer_norm_eps, mag_layer_norm_eps, beta_shift;
    float hidden_dropout, attn_dropout, mag_dropout;
    int dtype;       /* MB_DT_* */
    int max_batch, max_seq;
} mb_bert_config;

typedef struct mb_bert_engine mb_bert_engine;

int mb_bert_create(const mb_bert_config* cfg, mb_bert_engine** out);
void mb_bert_destroy(mb_bert_engine* e);
/* flat parameter layout (reference state-dict names).  decay != 0 -> weight_decay group of multimodal_driver.py:329-343 */
int mb_bert_num_tensors(const mb_bert_engine* e);
int mb_bert_tensor_info(const mb_bert_engine* e, int i, char* name, int name_cap, size_t* offset, size_t* numel,
                        int* ndim, int64_t* shape4, int* decay);
size_t mb_bert_param_count(const mb_bert_engine* e);   /* padded flat length (floats) */
size_t mb_bert_decay_count(const mb_bert_engine* e);   /* elements [0, n) are the weight-decay group */
void mb_bert_shadow_range(const mb_bert_engine* e, size_t* begin, size_t* end);   /* bf16 operand shadow range */
size_t mb_bert_workspace_bytes(const mb_bert_engine* e);
/* params / grads: flat fp32 [param_count]; shadow: bf16 [param_count] (dtype bf16) or NULL; ws: workspace. */
int mb_bert_bind(mb_bert_engine* e, float* params, float* grads, void* shadow, void* workspace, size_t ws_bytes);
/* refresh operand copies from the fp32 masters (bf16 shadow + packed MAG weights); call after loading weights. */
int mb_bert_sync_weights(mb_bert_engine* e, void* stream);

/* forward.  labels optional: num_labels == 1: fp32 [B] regression targets, fused MSE; num_labels > 1: fp32 [B] holding the class
 * index of each sample, fused cross entropy (bert.py:318-320) -> loss[0] (device, overwritten), loss_run += (optional).
 * training != 0 enables dropout keyed by (seed, step).  logits: fp32 [B][num_labels] (device, written). */
int mb_bert_forward(mb_bert_engine* e, const int64_t* input_ids, const float* visual, const float* acoustic,
                    const int64_t* attention_mask, const int64_t* token_type_ids, const float* labels, int B, int L,
                    int training, uint64_t seed, uint64_t step, float* logits, float* loss, float* loss_run,
                    void* stream);
/* backward of the last forward.  dlogits fp32 [B][num_labels], or NULL to use the fused MSE gradient
 * 2*(logit-label)/(B*num_labels)*loss_scale.  Gradients are ACCUMULATED into the bound flat grad buffer.
 * stage_begin/stage_end select a sub-range of [0, num_layers+2): 0 = head+pooler, 1..num_layers = encoder layers
 * (last layer first), num_layers+1 = MAG + embeddings -- lets the caller interleave gradient all-reduces. */
int mb_bert_backward(mb_bert_engine* e, const float* dlogits, const float* labels, float loss_scale, int stage_begin,
                     int stage_end, void* stream);
/* activations for API parity (device pointers into the workspace, valid until the next forward) */
const void* mb_bert_sequence_output(const mb_bert_engine* e);   /* [B*L][H] in dtype */
const float* mb_bert_pooled_output(const mb_bert_engine* e);    /* [B][H] fp32 (pre-dropout) */
/* Optional outputs of MAG_BertModel.forward (bert.py:147-156, 227-237).
 * hidden_state(i), i in [0, num_layers]: the encoder's all_hidden_states entry i -- i = 0 is the fused embedding MAG returns,
 * i = l + 1 the output of layer l -- [B*L][H] in dtype, valid until the next forward (they are the activations saved for
 * the backward anyway).  set_attention_output: the next forwards also write every layer's attention probabilities AFTER
 * dropout (what BertSelfAttention returns with output_attentions) to probs [num_layers][B][nh][L][L] fp32; NULL turns it off. */
const void* mb_bert_hidden_state(const mb_bert_engine* e, int i);
int mb_bert_set_attention_output(mb_bert_engine* e, float* probs);
/* Optional INPUTS of MAG_BertModel.forward (bert.py:160, 185-209).  Both are sticky until reset with NULL, apply to
 * mb_bert_forward / mb_bert_backward only (mb_bert_train_step returns MB_ERR_MODE while one is set), and point at caller-owned
 * device memory that must stay valid through the backward.
 * set_head_mask: fp32 [num_layers][num_heads]; the attention probabilities of head h in layer l are multiplied by
 *   head_mask[l][h] after dropout (BertSelfAttention; what get_head_mask broadcasts a 1-D or 2-D mask to).
 * set_inputs_embeds: fp32 [B*L][H] word embeddings used instead of the word_embeddings[input_ids] gather (input_ids may then
 *   be NULL); the word table receives no gradient, and after the backward inputs_embeds_grad returns the gradient of the
 *   given embeddings, fp32 [B*L][H] (workspace memory, valid until the next backward). */
int mb_bert_set_head_mask(mb_bert_engine* e, const float* head_mask);
int mb_bert_set_inputs_embeds(mb_bert_engine* e, const float* inputs_embeds);
/* position_ids (bert.py:211-216): int64 [B*L] device tensor of position-table rows for the next forwards / backwards, NULL =
 * BertEmbeddings' default arange(seq_len).  Sticky like head_mask; the single-call step refuses to run while it is set. */
int mb_bert_set_position_ids(mb_bert_engine* e, const int64_t* position_ids);
const float* mb_bert_inputs_embeds_grad(const mb_bert_engine* e);
/* Backward entry of the BASE model, for heads that live outside the engine: replaces stage 0 of mb_bert_backward.
 * d_sequence_output [B*L][H] (dtype; NULL = zero) is the gradient of outputs[0]; d_pooler_preact [B][H] (dtype; NULL = the
 * pooled output is unused) is the gradient of the pooler's PRE-activation, i.e. d_pooled * (1 - pooled^2) (pooled =
 * mb_bert_pooled_output).  Accumulates the pooler's weight / bias gradients; continue with mb_bert_backward(e, NULL, NULL,
 * 1.f, 1, num_layers + 2, stream). */
int mb_bert_backward_outputs(mb_bert_engine* e, const void* d_sequence_output, const void* d_pooler_preact, void* stream);
/* gradient buckets for data parallelism: range r of stage s covers flat elements [off, off+len) */
int mb_bert_stage_grad_ranges(const mb_bert_engine* e, int stage, size_t* offs, size_t* lens, int cap);

/* Known-zero gradients.  A step that runs the fused AdamW leaves the flat gradient buffer zeroed (the reference's
 * optimizer.zero_grad(), multimodal_driver.py:386); the next backward then STORES the layer weight gradients instead of adding
 * to them (no read of 28 MB per layer).  A host that zeroes the bound gradient buffer itself (model.zero_grad(), a stand-alone
 * mb_adamw_step with zero_grad) says so with known_zero = 1; anything else that writes gradients must leave / set it 0.  The
 * flag is consumed by the first stage of the next backward.  MB_WGRAD_OVERWRITE=0 disables the optimisation. */
int mb_bert_mark_grads_zero(mb_bert_engine* e, int known_zero);
/* Lazy zeroing.  The zeros mb_bert_train_step's AdamW would write over the layers' GEMM weight gradients are only ever
 * overwritten by the next backward, so a step that ends with the optimizer leaves that range "logically zero, physically stale"
 * (the rest of the buffer IS zeroed).  The engine writes the zeros itself before any of its own backwards that accumulates;
 * a host about to READ the bound gradient buffer after such a step (its own optimizer, a gradient exchange, a user looking at
 * .grad) calls mb_bert_materialize_grads first (a no-op when nothing is stale; mb_bert_grads_stale tells).  known_zero = 1
 * above also clears the state.  MB_ADAMW_KEEP=0 disables lazy zeroing: every fused step then really clears the buffer. */
int mb_bert_materialize_grads(mb_bert_engine* e, void* stream);
int mb_bert_grads_stale(const mb_bert_engine* e);

/* One whole optimizer step of train_epoch (multimodal_driver.py:354-388): `batch = tuple(t.to(DEVICE) ...)` staging, forward,
 * MSE (`:372-373`), loss.backward() (`:378`), optimizer.step() + optimizer.zero_grad() (`:384-386`) -- as TWO launches:
 *   1. a step prologue kernel that gathers the six batch tensors (device pointers, e.g. the landing buffer of an asynchronous
 *      H2D prefetch) into the engine's fixed staging buffers and writes this step's dropout keys (seed, step) and AdamW
 *      scalars (lr, bias-corrected step size from opt_step, grad_scale) into device memory;
 *   2. a replayed hipGraph with every other kernel of the step (captured on first use for each (B, L, output pointers)):
 *      one in-order kernel sequence -- the grouped weight-gradient launches run in line on the caller's stream.
 * Same kernels, same arithmetic, same dropout masks as mb_bert_forward + mb_bert_backward + 2 x mb_adamw_step with the same
 * (seed, step).  m, v: Adam moments parallel to the bound parameters; both NULL = no optimizer update (a gradient-accumulation
 * micro-step: gradients are accumulated, nothing is cleared).  With an update the gradient buffer is cleared in the same pass
 * (lazily over the layers' GEMM weights: see mb_bert_materialize_grads).
 * The two parameter groups of multimodal_driver.py:329-343 are [0, decay_count) with `weight_decay` and the rest with 0.
 * loss[0] = this step's MSE, loss_run (optional) += it.  mode: 1 = graph replay, 2 = the same sequence launched kernel by
 * kernel (reference for tests / profiling).  Pass the same logits / loss / m / v pointers every step: they are baked into the
 * captured graph (a new combination is captured again).  Not for data parallel runs: the gradient exchange is interleaved
 * with the backward stages from the host (mb_bert_backward + mb_bert_stage_grad_ranges). */
int mb_bert_train_step(mb_bert_engine* e, const int64_t* input_ids, const float* visual, const float* acoustic,
                       const int64_t* attention_mask, const int64_t* token_type_ids, const float* labels, int B, int L,
                       uint64_t seed, uint64_t step, float* logits, float* loss, float* loss_run, float* m, float* v, float lr,
                       float beta1, float beta2, float eps, float weight_decay, int opt_step, int correct_bias, float grad_scale,
                       float loss_scale, int mode, void* stream);
/* The same step cut at its backward stages, for data-parallel hosts that issue one gradient all-reduce piece per stage between
 * them (mb_bert_stage_grad_ranges says what each stage makes final): mb_bert_stage_forward = step prologue (batch gather from
 * device or pinned host pointers, this step's dropout keys) + forward + MSE as one replayed graph; mb_bert_stage_backward(stage),
 * stage = 0 .. num_layers + 1 in order, = that stage as one replayed graph.  mode 1 = graphs, 2 = the same kernels launched one by
 * one.  The optimizer update stays with the caller (mb_adamw_step).  Same arithmetic and dropout masks as mb_bert_forward +
 * mb_bert_backward with the same (seed, step). */
int mb_bert_stage_forward(mb_bert_engine* e, const int64_t* input_ids, const float* visual, const float* acoustic,
                          const int64_t* attention_mask, const int64_t* token_type_ids, const float* labels, int B, int L,
                          uint64_t seed, uint64_t step, float* logits, float* loss, float* loss_run, int mode, void* stream);
int mb_bert_stage_backward(mb_bert_engine* e, float loss_scale, int stage, int mode, void* stream);
/* the engine's staging copy of the last gathered input_ids ([B*L] int64, device): the rows of the word-embedding gradient */
const int64_t* mb_bert_staged_input_ids(const mb_bert_engine* e);
/* `batch = tuple(t.to(DEVICE) for t in batch)` (multimodal_driver.py:359, :396, :429) for callers that drive the passes
 * themselves (evaluation, data parallel): ONE gather launch copies the six batch tensors into the engine's staging buffers.
 * The sources may be pinned HOST memory (hipHostMalloc / torch pin_memory: the kernel reads it across PCIe, no copy engine, no
 * extra stream) or device memory.  labels may be NULL.  staged6 receives the device pointers to hand to mb_bert_forward /
 * mb_bert_backward in the order input_ids, visual, acoustic, attention_mask, token_type_ids, labels.  mb_bert_train_step
 * does this gather itself (its sources may be pinned host memory too). */
int mb_bert_load_batch(mb_bert_engine* e, const int64_t* input_ids, const float* visual, const float* acoustic,
                       const int64_t* attention_mask, const int64_t* token_type_ids, const float* labels, int B, int L,
                       const void** staged6, void* stream);
/* number of graphs captured / replays launched so far (tests, bench) */
int mb_bert_graph_stats(const mb_bert_engine* e, size_t* captures, size_t* launches);

/* Measurement hooks (bench.py): with profiling on, every per-layer grouped weight-gradient launch of mb_bert_backward is
 * bracketed by HIP timing events on the engine's internal side stream -- the stream that kernel runs on, which the
 * caller cannot see.  mb_bert_profile_wgrad_us waits for the last backward's events and returns the mean launch
 * duration over the layers, i.e. the in-step duration (concurrent with the dgrad chain), comparable with rocprofv3. */
int mb_bert_set_profiling(mb_bert_engine* e, int on);
int mb_bert_profile_wgrad_us(mb_bert_engine* e, float* avg_us);
/* the same for the two optimizer launches of the last mb_bert_train_step (first launch issued -> second complete, us) */
int mb_bert_profile_adamw_us(mb_bert_engine* e, float* us);

/* ------------------------------------------------------------------------------------------------ MAG-XLNet engine
 * MAG_XLNetForSequenceClassification forward / backward (xlnet.py:432-527 -> :15-429; XLNetLayer / SequenceSummary of
 * transformers 3.0.2) for the driver's configuration (bi-directional, no mems / perm_mask / target_mapping), L <= 128.
 * Same calling conventions as the mb_bert_* family.  Backward stages: 0 = summary + logits_proj, 1..n_layer = layers (last
 * first; the MAG backward runs inside the stage of layer `injection_index`), n_layer+1 = word embedding. */
typedef struct {
    int vocab_size, d_model, n_layer, n_head, d_inner, num_labels;
    int visual_dim, acoustic_dim, injection_index;
    float layer_norm_eps, mag_layer_norm_eps, beta_shift;
    float dropout, summary_last_dropout, mag_dropout;
    int dtype;
    int max_batch, max_seq;
} mb_xlnet_config;

typedef struct mb_xlnet_engine mb_xlnet_engine;

int mb_xlnet_create(const mb_xlnet_config* cfg, mb_xlnet_engine** out);
void mb_xlnet_destroy(mb_xlnet_engine* e);
int mb_xlnet_num_tensors(const mb_xlnet_engine* e);
/* decay: 1 = weight-decay group, 0 = no-decay group, 2 = frozen (transformer.mask_emb: no gradient in this configuration) */
int mb_xlnet_tensor_info(const mb_xlnet_engine* e, int i, char* name, int name_cap, size_t* offset, size_t* numel,
                         int* ndim, int64_t* shape4, int* decay);
size_t mb_xlnet_param_count(const mb_xlnet_engine* e);
size_t mb_xlnet_decay_count(const mb_xlnet_engine* e);
void mb_xlnet_shadow_range(const mb_xlnet_engine* e, size_t* begin, size_t* end);
size_t mb_xlnet_workspace_bytes(const mb_xlnet_engine* e);
int mb_xlnet_bind(mb_xlnet_engine* e, float* params, float* grads, void* shadow, void* workspace, size_t ws_bytes);
int mb_xlnet_sync_weights(mb_xlnet_engine* e, void* stream);
int mb_xlnet_forward(mb_xlnet_engine* e, const int64_t* input_ids, const float* visual, const float* acoustic,
                     const int64_t* attention_mask, const int64_t* token_type_ids, const float* labels, int B, int L,
                     int training, uint64_t seed, uint64_t step, float* logits, float* loss, float* loss_run,
                     void* stream);
int mb_xlnet_backward(mb_xlnet_engine* e, const float* dlogits, const float* labels, float loss_scale, int stage_begin,
                      int stage_end, void* stream);
const void* mb_xlnet_sequence_output(const mb_xlnet_engine* e);
/* input of layer i as XLNetModel collects it with output_hidden_states (xlnet.py:363-392; before the MAG injection), i = n_layer: the last output */
const void* mb_xlnet_hidden_state(const mb_xlnet_engine* e, int i);
/* output_attentions (xlnet.py:387-427): the softmax probabilities the last forward saved for its backward, layer `layer`:
 * [B * n_head][LP][LP] in the engine's dtype, LP = *padded_len = L rounded up to 32 (entries beyond L are zero), BEFORE the
 * attention dropout (the host multiplies the counter-hash mask of site XS_LAYER0 + 8 * layer in train mode).  Valid until the
 * next forward. */
const void* mb_xlnet_attention_probs(const mb_xlnet_engine* e, int layer, int* padded_len);
/* head_mask (xlnet.py:340-353, 383): fp32 [n_layer][n_head] device memory, sticky until reset with NULL; head h of layer l
 * contributes head_mask[l][h] times its attention output (attn_prob * head_mask after the dropout).  Explicit forwards /
 * backwards only: mb_xlnet_train_step returns MB_ERR_MODE while it is set. */
int mb_xlnet_set_head_mask(mb_xlnet_engine* e, const float* head_mask);
/* perm_mask / input_mask (xlnet.py:258-296): bytes [B][L][L] in device memory, sticky until reset with NULL; perm[b][i][j] != 0 <=>
 * data_mask[i, j, b] = input_mask[j, b] + perm_mask[i, j, b] > 0, i.e. query i may not attend to key j (i == j is always allowed:
 * non_tgt_mask, xlnet.py:288-296).  Combined with the attention_mask argument of the passes by OR.  Explicit forwards /
 * backwards only (mb_xlnet_train_step returns MB_ERR_MODE while it is set).  Masks the content stream (h) and, without the i == j
 * exemption, the query stream of mb_xlnet_query_stream (attn_mask_g, xlnet.py:288-296). */
int mb_xlnet_set_perm_mask(mb_xlnet_engine* e, const uint8_t* perm);
/* target_mapping -> the query stream g (xlnet.py:238-240, 306-313, 374-399; replaces the `g` half of XLNetLayer's two-stream
 * attention, transformers modeling_xlnet XLNetRelativeAttention.forward behind xlnet.py:374-385).  A post-pass over the last
 * mb_xlnet_forward, which must have been an EVAL pass (training == 0) without mems (else MB_ERR_MODE) and whose attention_mask /
 * token_type_ids buffers must still be alive: g starts as mask_emb on M rows per sample, every layer projects it with the
 * layer's q, maps it onto the L positions with target_mapping (fp32 [B][M][L], device; one-hot rows in the usual use, any weights
 * accepted), attends over that layer's content-stream keys / values / positions under the perm_mask / attention_mask WITHOUT the
 * self exemption, maps the result back to the M targets and shares post_attention + feed-forward with h.  head_mask / perm_mask
 * apply as set.  Nothing of the forward is overwritten (a backward after it is still valid; g itself has no backward here).
 * scratch: caller-owned device memory, 256-byte aligned, >= mb_xlnet_query_stream_scratch_bytes(e, B, M, L) for the forward's B, L.
 * On return its first n_layer + 1 blocks of mb_xlnet_query_stream_state_bytes(e, B, M) bytes each hold g in front of layer i
 * ([B][M][d_model], activation dtype; XLNetModel's hidden_states_g) -- block n_layer is output_g, what XLNetModel returns first when
 * target_mapping is given (xlnet.py:396-399; the final dropout is the identity in eval).  logits_g (fp32 [B][num_labels], device)
 * or NULL: SequenceSummary("last") + logits_proj on output_g's last row, as MAG_XLNetForSequenceClassification reads
 * transformer_outputs[0] (xlnet.py:506-509). */
size_t mb_xlnet_query_stream_scratch_bytes(const mb_xlnet_engine* e, int B, int M, int L);
size_t mb_xlnet_query_stream_state_bytes(const mb_xlnet_engine* e, int B, int M);
int mb_xlnet_query_stream(mb_xlnet_engine* e, const float* target_mapping, int M, void* scratch, size_t scratch_bytes,
                          float* logits_g, void* stream);
/* mems (xlnet.py:81-91, 244-245, 374-385: the hidden states cached from the previous segment; keys / values of layer l run over
 * cat([mems[l], h]), klen = mlen + qlen <= max_seq).  Explicit mb_xlnet_forward / mb_xlnet_backward passes only (the single-call
 * steps return MB_ERR_MODE while it is set).  The caller passes the segment as klen rows per sample whose first mlen rows are
 * placeholders (any ids, zero modalities, attention_mask 1, token_type 0): before layer l the engine replaces those rows of the
 * layer's input by mems[l] -- [n_layer][B][mlen][d_model] in the activation dtype, caller-owned device memory, NULL = none; the
 * backward clears the gradient of those rows at every layer seam (the memory is detached, xlnet.py:91) and keeps their share of the
 * k / v weight gradients. */
int mb_xlnet_set_mems(mb_xlnet_engine* e, const void* mems, int mlen);
int mb_xlnet_stage_grad_ranges(const mb_xlnet_engine* e, int stage, size_t* offs, size_t* lens, int cap);
int mb_xlnet_mark_grads_zero(mb_xlnet_engine* e, int known_zero);      /* as mb_bert_mark_grads_zero */
/* inputs_embeds (xlnet.py:306-313) / the base model's autograd edge (xlnet.py:396-405 returns autograd tensors), as for MAG-BERT:
 * mb_xlnet_set_inputs_embeds: fp32 [B*L][d_model] word embeddings for the next passes (NULL = input_ids again), their gradient
 * after a backward in mb_xlnet_inputs_embeds_grad; mb_xlnet_model_output: the last layer's output after the final dropout
 * (whole sequence, activation dtype, scratch valid until the next backward); mb_xlnet_backward_outputs: its gradient back in,
 * then stages 1 .. n_layer + 1 of mb_xlnet_backward. */
int mb_xlnet_set_inputs_embeds(mb_xlnet_engine* e, const float* inputs_embeds);
const float* mb_xlnet_inputs_embeds_grad(const mb_xlnet_engine* e);
const void* mb_xlnet_model_output(mb_xlnet_engine* e, void* stream);
int mb_xlnet_backward_outputs(mb_xlnet_engine* e, const void* d_output, void* stream);
int mb_xlnet_set_profiling(mb_xlnet_engine* e, int on);                 /* as mb_bert_set_profiling (the grouped launch has 7 problems) */
int mb_xlnet_profile_wgrad_us(mb_xlnet_engine* e, float* avg_us);
int mb_xlnet_profile_adamw_us(mb_xlnet_engine* e, float* us);
int mb_xlnet_materialize_grads(mb_xlnet_engine* e, void* stream);       /* as mb_bert_materialize_grads */
int mb_xlnet_grads_stale(const mb_xlnet_engine* e);
/* the MAG-XLNet counterparts of mb_bert_train_step / mb_bert_load_batch / mb_bert_graph_stats (same contracts; one iteration of
 * train_epoch, multimodal_driver.py:359-386, for the xlnet-base-cased model).  The two parameter groups are [0, decay_count) and
 * [decay_count, trainable_count); the frozen transformer.mask_emb slot behind them is never updated (HF AdamW skips grad-less
 * parameters).  MB_ERR_MODE when the engine was created with MB_OVERLAP_WGRAD=1 (side-stream weight gradients). */
int mb_xlnet_train_step(mb_xlnet_engine* e, const int64_t* input_ids, const float* visual, const float* acoustic,
                        const int64_t* attention_mask, const int64_t* token_type_ids, const float* labels, int B, int L,
                        uint64_t seed, uint64_t step, float* logits, float* loss, float* loss_run, float* m, float* v, float lr,
                        float beta1, float beta2, float eps, float weight_decay, int opt_step, int correct_bias, float grad_scale,
                        float loss_scale, int mode, void* stream);
int mb_xlnet_load_batch(mb_xlnet_engine* e, const int64_t* input_ids, const float* visual, const float* acoustic,
                        const int64_t* attention_mask, const int64_t* token_type_ids, const float* labels, int B, int L,
                        const void** staged6, void* stream);
int mb_xlnet_graph_stats(const mb_xlnet_engine* e, size_t* captures, size_t* launches);
size_t mb_xlnet_trainable_count(const mb_xlnet_engine* e);

/* ------------------------------------------------------------------------------------------------ data parallel (new)
 * The reference is single-device (global_configs.py:4,7; multimodal_driver.py:21 imports a DistributedSampler it never uses).
 * Data-parallel fine-tuning here = one process per GPU, replicated parameters, the minibatch cut per rank, and ONE logical
 * all-reduce(sum) of the flat fp32 gradient buffer per optimizer step, issued from C on a side HIP stream in pieces as the backward
 * finishes them (csrc/comm.hip).  An mb_comm owns that stream, its events and the backend:
 *   mb_comm_create_rccl      -- RCCL (dlopen'ed at run time; a process that already carries an RCCL, e.g. PyTorch's, shares it).
 *                               id128 = the 128-byte ncclUniqueId rank 0 got from mb_comm_unique_id and handed to every rank
 *                               (over torch.distributed, a file, MPI: the caller's plumbing);
 *   mb_comm_create_callbacks -- host callbacks (tests: two gloo ranks on one GPU).  all_reduce(ctx, buf, count, dtype, stream):
 *                               in-place sum over the ranks of `count` elements of MB_DT_*; all_gather(ctx, buf, bytes_per_rank,
 *                               stream): in place, rank r's piece at buf + r * bytes_per_rank; both ordered on `stream`.
 * Scratch (caller-owned device memory, mb_comm_bind_scratch): the bf16 wire staging ([n_params] bf16, only with wire_dtype =
 * MB_DT_BF16) and the buffers of the row-wise word-embedding exchange (vocab x world slot table, world x capacity_rows ids and
 * fp32 rows; vocab = 0 -> the table travels in the dense tail piece).  mb_comm_exposed_ms: with timing on, how long the compute
 * stream of the LAST step was stalled on the exchange (synchronises on the timing events).  Error text: mb_comm_last_error(). */
typedef struct mb_comm mb_comm;
typedef int (*mb_all_reduce_cb)(void* ctx, void* buf, size_t count, int dtype, void* stream);
typedef int (*mb_all_gather_cb)(void* ctx, void* buf, size_t bytes_per_rank, void* stream);
int mb_comm_unique_id(void* id128);
int mb_comm_create_rccl(const void* id128, int rank, int world, mb_comm** out);
int mb_comm_create_callbacks(int rank, int world, mb_all_reduce_cb all_reduce, mb_all_gather_cb all_gather, void* ctx, mb_comm** out);
void mb_comm_destroy(mb_comm* c);
int mb_comm_rank(const mb_comm* c);
int mb_comm_world(const mb_comm* c);
void* mb_comm_stream(const mb_comm* c);                 /* the comm stream (hipStream_t) */
size_t mb_comm_scratch_bytes(int world, int wire_dtype, size_t n_params, int vocab, int hidden, int capacity_rows);
int mb_comm_bind_scratch(mb_comm* c, void* scratch, size_t bytes, int wire_dtype, size_t n_params, int vocab, int hidden, int capacity_rows);
/* stand-alone pieces of the exchange (tests; both in place, ordered on `stream`): sum of buf[0, count) over the ranks in the wire
 * format; row-wise sum of a [vocab][hidden] fp32 table of which this rank touched the rows ids[0, T) (T <= capacity_rows) */
int mb_comm_all_reduce(mb_comm* c, float* buf, size_t count, void* stream);
int mb_comm_exchange_rows(mb_comm* c, float* table, const int64_t* ids, int T, void* stream);
/* Gradient accumulation (/root/reference/multimodal_driver.py:375-376, 383-386): micro-steps are plain mb_*_train_step calls without
 * m / v (they accumulate and exchange nothing); the mb_*_train_step_dp that follows them must move the word-embedding table DENSELY
 * (its gradient holds the rows of every micro-step, not only this step's ids): mb_comm_set_row_exchange(c, 0) before it, 1 after. */
int mb_comm_set_row_exchange(mb_comm* c, int rowwise);
/* Sharded optimizer update (ZeRO-1 over the layers' GEMM weights; NEW: the reference runs one AdamW over everything,
 * multimodal_driver.py:345, 384-386).  With sharding on, mb_*_train_step_dp reduce-scatters every piece of layer GEMM-weight
 * gradients instead of all-reducing it, updates this rank's slice of every piece only, and all-gathers what the next forward reads
 * (the bf16 shadow in bf16 mode, the fp32 parameters in fp32 mode) on the comm stream.  With more than one piece the piece finished
 * last (the lowest layers, needed first) stays replicated and the NEXT mb_*_train_step_dp runs its forward as one graph per piece,
 * each waiting for the gather of its own piece only; any other consumer of the weights calls mb_comm_join(c, its stream) first.  In bf16 mode the fp32 masters -- and in either
 * mode Adam's m / v -- of the other ranks' slices are stale until mb_comm_gather_shards(c, flat buffer, 4, stream) refreshes them
 * (state_dict / checkpoints).  mb_comm_shard_slices: this rank's [begin, end) slices of the last sharded step. */
int mb_comm_set_sharding(mb_comm* c, int on);
int mb_comm_sharding(const mb_comm* c);
int mb_comm_join(mb_comm* c, void* stream);
int mb_comm_gather_shards(mb_comm* c, void* base, int elem_bytes, void* stream);
int mb_comm_shard_slices(const mb_comm* c, size_t* begin_end_pairs, int max_pairs);
int mb_comm_set_timing(mb_comm* c, int on);
int mb_comm_exposed_ms(mb_comm* c, float* ms);
int mb_comm_stats(const mb_comm* c, size_t* pieces, size_t* bytes);      /* collectives issued / bytes handed to them in the last step */
const char* mb_comm_last_error(void);
/* One optimizer step of a data-parallel rank as ONE engine call: mb_bert_train_step with the exchange inside.  The step runs as a
 * chain of LINEAR replayed graphs (a graph with a cross-stream fork replays on ROCm 7.2's slow path): forward + head + the
 * backward of the top 4 layers | 4 more | 2 more | the last 2 layers + MAG + embeddings | AdamW of the GEMM weights of the ten
 * layers reduced early | AdamW of the rest (MB_DP_CHUNKS="4,4,2,2" / MB_DP_CHUNK=n change the cut).  Between two of them the host
 * records an event and issues that segment's all-reduce on the comm stream; what the last backward segment produces -- its two
 * layers and the tail (everything that is not a layer's GEMM weight; the word-embedding table row-wise) -- travels under the first
 * AdamW launch.  Everything the single-call step has stays: the deferred LayerNorm reduction, stored (not accumulated) weight
 * gradients, lazy zeroing.  grad_scale = 1 / world for the mean over the global batch (the pieces are SUMs).  m, v must be given
 * (a gradient-accumulation micro-step has nothing to exchange: use mb_bert_train_step). */
int mb_bert_train_step_dp(mb_bert_engine* e, const int64_t* input_ids, const float* visual, const float* acoustic,
                          const int64_t* attention_mask, const int64_t* token_type_ids, const float* labels, int B, int L,
                          uint64_t seed, uint64_t step, float* logits, float* loss, float* loss_run, float* m, float* v, float lr,
                          float beta1, float beta2, float eps, float weight_decay, int opt_step, int correct_bias, float grad_scale,
                          float loss_scale, int mode, void* stream, mb_comm* comm);
int mb_xlnet_train_step_dp(mb_xlnet_engine* e, const int64_t* input_ids, const float* visual, const float* acoustic,
                           const int64_t* attention_mask, const int64_t* token_type_ids, const float* labels, int B, int L,
                           uint64_t seed, uint64_t step, float* logits, float* loss, float* loss_run, float* m, float* v, float lr,
                           float beta1, float beta2, float eps, float weight_decay, int opt_step, int correct_bias, float grad_scale,
                           float loss_scale, int mode, void* stream, mb_comm* comm);

#ifdef __cplusplus
}
#endif
#endif
