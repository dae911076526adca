/* mmdti_hip.h -- C ABI of libmmdti_hip.so: the MI355X (gfx950) kernels behind the
 * MM-DTI dual-encoder contrastive fine-tune step.
 *
 * The reference (ndlongvn/MM-DTI) has no native layer: its hot path is Python
 * calling torch / Uni-Core / HuggingFace ops.  Each entry point below replaces
 * the ATen / Uni-Core-CUDA kernels that one reference call site reaches; the
 * call site is cited per function as file:line relative to the reference root.
 * The host-side mirror of the reference's module interface (same class names,
 * ctor/forward signatures, parameter names) lives in mm-dti_amd/mmdti_hip/models
 * and binds these symbols through ctypes.
 *
 * Conventions
 *   - plain pointers + sizes only; every pointer is DEVICE memory unless noted.
 *   - `stream` is a hipStream_t passed as void*; every call only enqueues work on
 *     it (no host sync, no allocation, graph-capture safe).
 *   - bf16 tensors are raw uint16 bit patterns (`void*` in the signature).
 *   - return value: MMDTI_OK or an error code; mmdti_last_error() gives the text
 *     (thread-local).  Argument validation happens on the host BEFORE any launch.
 *   - "atomic accumulate" outputs must be zeroed by the caller (gradient arena).
 *   - dropout: p in [0,1); mask = counter-based hash RNG(seed, site, element index); the
 *     backward entry regenerates the same mask from (seed, site).
 */
#ifndef MMDTI_HIP_H
#define MMDTI_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

typedef void* mmdti_stream_t;

#define MMDTI_OK 0
#define MMDTI_ERR_INVALID 1
#define MMDTI_ERR_LAUNCH 2

#define MMDTI_ACT_NONE 0
#define MMDTI_ACT_GELU 1      /* erf GELU; optional aux_out = pre-activation (bf16) */
#define MMDTI_ACT_GELU_BWD 2  /* multiply by gelu'(aux_in) */
#define MMDTI_ACT_GELU_G 4    /* erf GELU; aux_out (required) = gelu'(pre-activation) as bf16: the forward has the erf and
                                 the Gaussian in hand, so the backward (MUL_AUX) is one multiply per element */
#define MMDTI_ACT_MUL_AUX 5   /* multiply by aux_in (bf16) */

#define MMDTI_DT_F32 0
#define MMDTI_DT_BF16 1
#define MMDTI_DT_F32_ATOMIC 2 /* atomicAdd into fp32 C (split-K / gradient accumulation) */
/* fp16 forward operands (the default precision mode since round 4; the reference's own AMP dtype, tasks/trainer.py:181-182 -- three
 * more mantissa bits than bf16 at the same matrix-pipe rate; profiles/r03_rounding_sites_fp16.json): */
#define MMDTI_DT_F16 3        /* the 16-bit output is fp16 instead of bf16 */
#define MMDTI_DT_AB_F16 16    /* OR-ed into mmdti_gemm_bf16's c_dtype: A and B hold fp16 (forward Linear shapes only: row-major A,
                                 weight-layout B, no split-K, no batch) */
#define MMDTI_DT_B_F16 32     /* OR-ed into mmdti_gemm_bf16's c_dtype: B holds fp16 beside a bf16 A -- the weight gradient dW += dy^T.x
                                 (transA = transB = 1, MMDTI_DT_F32_ATOMIC) whose x is a forward activation saved as fp16.  The tile is
                                 fetched as it lies in memory and converted to bf16 (round to nearest even: the values
                                 mmdti_cast_f16_bf16 would write) between LDS and the matrix pipe; no pass over HBM */

#define MMDTI_CT_REGRESS 0
#define MMDTI_CT_SINGLE 1
#define MMDTI_CT_MULTI 2

const char* mmdti_last_error(void);
int mmdti_abi_version(void);
/* Run-time tuning switches (A/B measurements inside one process; defaults come from the environment variable of the same
 * upper-cased name, e.g. MMDTI_GEMM_BIG).  Known names: "gemm_big" (0 off, 1 where the shape fills the chip, 2 every
 * eligible shape); "gemm_dbg" (measurement only: 1 = the large-tile GEMM skips its epilogue); "gemm_small" (1: 64 x 64 tiles for launches of
 * few tiles -- small batches; 0: 128 x 128 everywhere) and "gemm_deep" (1: four-stage LDS-DMA ring for launches of at most one
 * workgroup per CU); all paths give bit-identical results.  Returns MMDTI_ERR_INVALID for an unknown name.  Not part of any reference interface. */
int mmdti_set_option(const char* name, int value);

/* ---- GEMM: C = epi(alpha * A.B^T) ----------------------------------------------------------
 * Replaces nn.Linear / torch.bmm fwd+bwd: unicore in_proj/out_proj/fc1/fc2 (models/transformers.py:137-139),
 * gbf_proj (models/mm_model.py:554), RobertaModel linears (mm_model.py:562), InfoNCE projections
 * (models/infonce.py:28-29), BertCrossEncoder linears (models/mm_module.py:486-488,527,545,581).
 * A: transA=0 -> [M,K] row-major (lda), transA=1 -> stored [K,M].  B: transB=0 -> [N,K] ("weight" layout),
 * transB=1 -> stored [K,N].  lda/ldb multiples of 8, 16-byte aligned bases.  If K%8 != 0 the k-contiguous
 * operand must hold zeros in its padded tail.  Batch index z = outer*batch_inner + inner.
 * colsum_out (nullable, [N] fp32, +=): column sums of the stored C -- the bias gradient of the Linear whose output
 * gradient C is -- accumulated by the epilogue (aligned, unbatched, unsplit outputs only).
 * arowsum_out (nullable, [M] fp32, +=, transA only, unbatched): sum over k of op(A)[m][k].  For a weight gradient
 * dW = dy^T.x (A = dy stored [tokens, out]) that is the Linear's bias gradient, taken inside the same pass over dy.
 * workspace (nullable, workspace_bytes): device scratch for split-K.  With it, a split-K product of a large-tile shape
 * writes one fp32 partial per split with plain stores and a second pass adds their sum into C -- global float atomics
 * run at a fifth of the store rate on this chip -- instead of atomicAdd-ing every partial into C.  Without it (or when
 * it is too small: splits * M * N * 4 bytes) the atomic form runs.  Contents on return are unspecified. */
int mmdti_gemm_bf16(mmdti_stream_t stream, const void* A, const void* B, void* C, int M, int N, int K, int lda,
                    int ldb, int ldc, int transA, int transB, int batch_outer, int batch_inner, long long sAo,
                    long long sAi, long long sBo, long long sBi, long long sCo, long long sCi, int splitk,
                    float alpha, float beta, const float* bias, const float* residual, int ldr, int act,
                    const void* aux_in, void* aux_out, int ld_aux, int c_dtype, float drop_p,
                    unsigned long long seed, unsigned int site, float* colsum_out, float* arowsum_out, void* workspace,
                    long long workspace_bytes);

/* ---- Linear + residual + LayerNorm in one kernel ------------------------------------------------
 * The Linear that closes a residual branch and the LayerNorm that follows it: unicore out_proj -> final_layer_norm and fc2 -> the
 * next layer's self_attn_layer_norm / the encoder's final_layer_norm (pre-LN, models/transformers.py:137-139,160-161); HF / BertCross
 * attention.output.dense -> LayerNorm and output.dense -> LayerNorm (post-LN, mm_model.py:562, mm_module.py:523-534,561-587):
 *   x_out[M,N] (fp32) = residual + dropout(A.W^T + bias);   h = (x - mean) * rstd * gamma + beta  -> ln_f32 and / or ln_bf16
 * (either may be null); mean, rstd: [M] fp32 (row mean and 1/sqrt(biased var + eps), what mmdti_layernorm_bwd takes).
 * A: [M,K] bf16 (lda), W: [N,K] bf16 (ldb); N == 512 (a workgroup owns whole rows), K % 64 == 0.  Dropout counters are those of
 * mmdti_gemm_bf16's epilogue (element row*N + col), so the fused and the unfused path draw the same mask. */
int mmdti_gemm_ln_bf16(mmdti_stream_t stream, const void* A_bf16, const void* W_bf16, const float* bias, const float* residual, int M,
                       int N, int K, int lda, int ldb, int ldr, float drop_p, unsigned long long seed, unsigned int site, float* x_out,
                       const float* gamma, const float* beta, float eps, float* ln_f32, void* ln_bf16, float* mean, float* rstd,
                       int f16 /* bit 0: A and W hold fp16; bit 1: ln_bf16 receives fp16 */);

/* ---- Grouped weight gradients of one transformer layer -----------------------------------------
 * The weight half of nn.Linear's backward for up to 8 Linears that saw the SAME token rows (unicore in_proj / out_proj /
 * fc1 / fc2 of one encoder layer, models/transformers.py:137-139; query|key|value / dense / intermediate / output of one
 * RoBERTa or BertCrossAttention layer, mm_model.py:562, mm_module.py:470-587):
 *   dw[i] [n_out[i], lddw[i]] fp32 += dy[i]^T . x[i],   db[i] [n_out[i]] fp32 += column sums of dy[i]   (db or db[i] may be null)
 * dy[i]: [rows, ldy[i]] bf16, x[i]: [rows, ldx[i]] bf16; n_out, n_in multiples of 256; rows a multiple of 64.
 * One launch over all output tiles with a K split chosen to fill the chip; partial sums go through `workspace`
 * (mmdti_linear_dw_grouped_splits(total 256x256 tiles, rows) * sum_i n_out[i]*n_in[i] * 4 bytes) and a second pass adds
 * them into dw.  The argument tables are HOST arrays of nprob entries. */
int mmdti_linear_dw_grouped(mmdti_stream_t stream, int nprob, const void* const* dy_bf16, const void* const* x_bf16,
                            float* const* dw, float* const* db, const int* n_out, const int* n_in, const int* ldy,
                            const int* ldx, const int* lddw, int rows, void* workspace, long long workspace_bytes,
                            int x_f16 /* 1: every x[i] holds fp16 (saved forward activations of the fp16 forward-operand mode),
                                         converted to bf16 between LDS and the matrix pipe -- see MMDTI_DT_B_F16 */);
int mmdti_linear_dw_grouped_splits(int tiles, int rows);

/* ---- one Uni-Mol encoder layer's forward / backward behind one call each (launch sequencing in the library: at 16-32 molecules
 * the Python side of ~500 launches per step sets the pace).  Replaces the per-layer body of the encoder forward / backward -- unicore
 * TransformerEncoderLayer as driven by models/transformers.py:136-139 -- with the SAME eight launches the op-by-op host path
 * issues (fc2 / fc1 input gradients, LayerNorm-2 backward, out_proj input gradient, pair-attention backward, in_proj input
 * gradient, LayerNorm-1 backward, grouped weight gradients): bit-identical results.  Shapes and the workspace layout: layers.hip. */
int mmdti_unimol_layer_fwd(mmdti_stream_t stream, int M, int B, int N, int H, int D, int F, int ld, float scale, float p_res,
                           float p_att, unsigned long long seed, unsigned int site_att, unsigned int site_o, unsigned int site_f,
                           const float* x, const void* h1, const void* s_in, const unsigned char* key_pad, int pair_layout,
                           const int* key_tiles, int rag_store, const int* row_off, const void* w_in, const float* b_in,
                           const void* w_out, const float* b_out, const float* g_ln2, const float* bt_ln2, float eps2,
                           const void* w_fc1, const float* b_fc1, int act_fwd, const void* w_fc2, const float* b_fc2,
                           int next_mode, const float* g_next, const float* bt_next, float eps_next, int ln_max_k, void* qkv,
                           void* s_out, void* o_att, float* x1, void* h2, float* m2, float* r2, void* u_aux, void* a_act,
                           float* x_out, void* ln_out, float* mn, float* rn,
                           int fwd_f16 /* 1: the fp16 forward-operand mode -- every 16-bit operand of a forward GEMM (h1, weights, q | k | v,
                                          o_att, h2, a_act, a 16-bit ln_out) holds fp16; pair_layout 3 only */);
int mmdti_unimol_layer_bwd(mmdti_stream_t stream, int M, int B, int N, int H, int D, int F, int ld, float scale,
                           float p_res, float p_att, unsigned long long seed, unsigned int site_f_below, unsigned int site_o,
                           unsigned int site_att, const float* dx_in, const void* dy2, float* dx_out, void* dx16_out,
                           float* db_below, const void* a_act, const void* u_aux, int act_dx, const void* h2, const float* x1,
                           const float* m2, const float* r2, const void* o_att, const void* qkv, const void* s_logits,
                           const void* h1, const float* x0, const float* m1, const float* r1, const void* w_fc2,
                           const void* w_fc1, const void* w_out, const void* w_in, const float* g_ln2, const float* g_ln1,
                           float* dw_fc2, float* dw_fc1, float* dw_out, float* dw_in, float* db_fc1, float* db_out,
                           float* db_in, float* dg_ln2, float* dbt_ln2, float* dg_ln1, float* dbt_ln1, void* G,
                           int pair_layout, int g_in_zero, const int* key_tiles, const int* row_off, void* ws,
                           long long ws_bytes, int fwd_f16 /* 1: the saved a_act, h2, o_att, h1, qkv hold fp16 (weights: the bf16 shadow) */);

/* ---- ALL layers of the Uni-Mol encoder behind one call per direction (the loop of models/transformers.py:136-139 in :96-183).  At
 * the reference's 16-32 molecules even the per-layer calls leave ~90 us of Python per layer and direction; these take the layers'
 * parameters as pointer tables and keep every saved tensor at a fixed offset of ONE caller-owned arena.  The launches are those of
 * nl x mmdti_unimol_layer_fwd / _bwd: bit-identical results.
 *   mmdti_unimol_stack_layout: out[0] = arena bytes per layer, out[1] = bytes of the backward workspace (s_bytes: one layer's pair
 *     logits; dw_slab_bytes: split-K slabs of one layer's grouped weight gradients, see mmdti_linear_dw_grouped_splits).
 *   params [nl][12]: w_in, b_in, w_out, b_out, g_ln2, bt_ln2, w_fc1, b_fc1, w_fc2, b_fc2, g_ln1, bt_ln1 (16-bit forward weights).
 *   bparams [nl][6]: w_fc2, w_fc1, w_out, w_in (bf16), g_ln2, g_ln1.   grads [nl][12]: dw_fc2, dw_fc1, dw_out, dw_in, db_fc2, db_fc1,
 *     db_out, db_in, dg_ln2, dbt_ln2, dg_ln1, dbt_ln1 (fp32, accumulated).
 *   x0 / h1_0 / m1_0 / r1_0: the stream entering layer 0 and its LayerNorm-1 output + statistics (the caller's); the last layer writes
 *   x_last, s_last and (g_final non-null) the final LayerNorm out_final / mean_final / rstd_final.  Dropout sites: site0 + 3 l + {0, 1, 2}.
 *   dw_stream + events (hipEvent_t [3]; both null: everything on `stream`): the grouped weight gradients of a layer run on dw_stream
 *   under the layer below; `stream` has joined dw_stream when mmdti_unimol_stack_bwd returns. */
int mmdti_unimol_stack_layout(int M, int D, int F, long long s_bytes, long long dw_slab_bytes, long long* out);
int mmdti_unimol_stack_fwd(mmdti_stream_t stream, int nl, int M, int B, int N, int H, int D, int F, int ld, float scale, float p_res,
                           float p_att, unsigned long long seed, unsigned int site0, const float* x0, const void* h1_0,
                           const void* s_in, const unsigned char* key_pad, int pair_layout, const int* key_tiles,
                           int rag_store_last, const int* row_off, const void* const* params, int act_fwd, float eps_ln,
                           const float* g_final, const float* bt_final, float eps_final, int ln_max_k, void* arena,
                           long long arena_bytes, long long s_bytes, float* x_last, void* s_last, float* out_final,
                           float* mean_final, float* rstd_final, int fwd_f16);
int mmdti_unimol_stack_bwd(mmdti_stream_t stream, int nl, int M, int B, int N, int H, int D, int F, int ld, float scale, float p_res,
                           float p_att, unsigned long long seed, unsigned int site0, const float* dx_in, const void* dy2_in,
                           float* dx_final, const float* x0, const void* h1_0, const float* m1_0, const float* r1_0,
                           const void* s_last, const void* const* bparams, int act_dx, void* const* grads, void* G, int pair_layout,
                           int g_first_zero, const int* key_tiles, const int* row_off, const void* arena, long long arena_bytes,
                           long long s_bytes, void* ws, long long ws_bytes, long long dw_slab_bytes, int fwd_f16,
                           mmdti_stream_t dw_stream, void* const* events);

/* The same for one post-LN BERT layer with self-attention, fused q | k | v projection and the fused attention kernels (HF RobertaLayer
 * reached from models/mm_model.py:562): six launches forward, eight backward.  Shapes and the workspace layout: layers.hip. */
int mmdti_bert_layer_fwd(mmdti_stream_t stream, int Mq, int B, int L, int heads, int D, int F, float scale, float p_hid, float p_att,
                         unsigned long long seed, unsigned int site_att, unsigned int site_o, unsigned int site_f,
                         const float* s1_32, const void* s1_16, const float* key_add, const int* q_off, const int* k_off,
                         const int* k_cnt, int q_rows, const void* w_qkv, const float* b_qkv, const void* w_o, const float* b_o,
                         const float* g_ln1, const float* bt_ln1, const void* w_i, const float* b_i, int act_fwd,
                         const void* w_o2, const float* b_o2, const float* g_ln2, const float* bt_ln2, float eps, int ln_max_k,
                         void* qkv, void* ctx, float* stats, float* y, float* a32, void* a16, float* am, float* ar, void* u_aux,
                         void* i_act, float* z, float* out32, void* out16, float* zm, float* zr,
                         int fwd_f16 /* 1: s1_16, the weights, ctx, a16, i_act, out16 hold fp16 (q | k | v stay bf16) */);
int mmdti_bert_layer_bwd(mmdti_stream_t stream, int Mq, int B, int L, int heads, int D, int F, float scale, float p_hid, float p_att,
                         unsigned long long seed, unsigned int site_att, unsigned int site_o, unsigned int site_f,
                         const float* dout, float* ds1, const void* s1_16, const float* key_add, const int* q_off,
                         const int* k_off, const int* k_cnt, int q_rows, const void* qkv, const void* ctx, const float* stats,
                         const float* y, const void* a16, const float* am, const float* ar, const void* u_aux, int act_dx,
                         const void* i_act, const float* z, const float* zm, const float* zr, const void* w_qkv,
                         const void* w_o, const void* w_i, const void* w_o2, const float* g_ln1, const float* g_ln2,
                         float* dw_qkv, int lddw_qkv, float* db_qkv, float* dw_o, float* db_o, float* dw_i, float* db_i,
                         float* dw_o2, float* db_o2, float* dg_ln1, float* dbt_ln1, float* dg_ln2, float* dbt_ln2, void* ws,
                         long long ws_bytes, int fwd_f16 /* 1: the saved s1_16, ctx, a16, i_act hold fp16 (weights: the bf16 shadow) */);
/* The CROSS-attention variant (BertCrossAttentionLayer, mm_module.py:615-626 through :663-677: queries from s1 [Mq rows], keys and values
 * from s2 [Mk rows]): the six forward launches behind one call (query projection, fused key | value projection, fused attention, the two
 * closers, intermediate + GELU), and the backward UP TO the weight gradients (ten launches; ds1 / ds2: the gradients of the two inputs).
 * The backward's five bf16 activation gradients (dzb, du, dyb, dq, dkv) are the caller's: they are the A operands of the layer's weight
 * gradients, which the caller launches (mmdti_linear_dw_grouped takes one token-row count per launch, this layer has two).
 * ws of the backward: da [Mq,D] | dctx [Mq,D] (bf16) | dz [Mq,D] f32 | the attention backward's row term. */
int mmdti_bert_cross_layer_fwd(mmdti_stream_t stream, int Mq, int Mk, int B, int Lq, int Lk, int heads, int D, int F, float scale,
                               float p_hid, float p_att, unsigned long long seed, unsigned int site_att, unsigned int site_o,
                               unsigned int site_f, const float* s1_32, const void* s1_16, const void* s2_16, const float* key_add,
                               const int* q_off, const int* k_off, const int* k_cnt, int q_rows, const void* w_q, const float* b_q,
                               const void* w_kv, const float* b_kv, const void* w_o, const float* b_o, const float* g_ln1,
                               const float* bt_ln1, const void* w_i, const float* b_i, int act_fwd, const void* w_o2,
                               const float* b_o2, const float* g_ln2, const float* bt_ln2, float eps, int ln_max_k, void* q,
                               void* kv, void* ctx, float* stats, float* y, float* a32, void* a16, float* am, float* ar,
                               void* u_aux, void* i_act, float* z, float* out32, void* out16, float* zm, float* zr, int fwd_f16);
int mmdti_bert_cross_layer_bwd(mmdti_stream_t stream, int Mq, int Mk, int B, int Lq, int Lk, int heads, int D, int F, float scale,
                               float p_hid, float p_att, unsigned long long seed, unsigned int site_att, unsigned int site_o,
                               unsigned int site_f, const float* dout, float* ds1, float* ds2, const float* key_add,
                               const int* q_off, const int* k_off, const int* k_cnt, int q_rows, const void* q, const void* kv,
                               const float* stats, const float* y, const float* am, const float* ar, const void* u_aux,
                               int act_dx, const float* z, const float* zm, const float* zr, const void* w_q, const void* w_kv,
                               const void* w_o, const void* w_i, const void* w_o2, const float* g_ln1, const float* g_ln2,
                               float* db_o, float* db_o2, float* dg_ln1, float* dbt_ln1, float* dg_ln2, float* dbt_ln2, void* dzb,
                               void* du, void* dyb, void* dq, void* dkv, void* ws, long long ws_bytes);
/* ---- ALL layers of tower 2 behind one call per direction (HF RobertaEncoder's layer loop, reached from models/mm_model.py:562): as
 * the Uni-Mol stack above -- pointer tables for the parameters, one activation arena, nl x mmdti_bert_layer_fwd / _bwd, bit-identical.
 *   mmdti_bert_stack_layout: out[0] = arena bytes per layer, out[1] = backward workspace bytes (stats_bytes: one layer's softmax
 *     statistics; nrow: heads * q_rows packed, B * heads * L dense; dw_slab_bytes: split-K slabs of one layer's weight gradients).
 *   params [nl][12]: w_qkv, b_qkv, w_o, b_o, g_ln1, bt_ln1, w_i, b_i, w_o2, b_o2, g_ln2, bt_ln2 (16-bit forward weights, q | k | v fused).
 *   bparams [nl][6]: w_qkv, w_o, w_i, w_o2 (bf16), g_ln1, g_ln2.   grads [nl][12]: dw_qkv, db_qkv, dw_o, db_o, dw_i, db_i, dw_o2, db_o2,
 *     dg_ln1, dbt_ln1, dg_ln2, dbt_ln2 (fp32, accumulated).  Dropout sites: site0 + 3 l + {0: attention, 1: output.dense, 2: FFN}. */
int mmdti_bert_stack_layout(int Mq, int D, int F, long long stats_bytes, long long nrow, long long dw_slab_bytes, long long* out);
int mmdti_bert_stack_fwd(mmdti_stream_t stream, int nl, int Mq, int B, int L, int heads, int D, int F, float scale, float p_hid,
                         float p_att, unsigned long long seed, unsigned int site0, const float* s1_32_0, const void* s1_16_0,
                         const float* key_add, const int* q_off, const int* k_off, const int* k_cnt, int q_rows,
                         const void* const* params, int act_fwd, float eps, int ln_max_k, void* arena, long long arena_bytes,
                         long long stats_bytes, float* out32_last, int fwd_f16);
int mmdti_bert_stack_bwd(mmdti_stream_t stream, int nl, int Mq, int B, int L, int heads, int D, int F, float scale, float p_hid,
                         float p_att, unsigned long long seed, unsigned int site0, const float* dout, float* ds1_final,
                         const void* s1_16_0, const float* key_add, const int* q_off, const int* k_off, const int* k_cnt,
                         int q_rows, const void* const* bparams, int act_dx, void* const* grads, int lddw_qkv, const void* arena,
                         long long arena_bytes, long long stats_bytes, void* ws, long long ws_bytes, long long dw_slab_bytes,
                         int fwd_f16);


/* ---- LayerNorm (unicore LayerNorm eps 1e-5: transformers.py:69,71,114,161; BertLayerNorm eps 1e-12:
 * mm_module.py:320-333; HF nn.LayerNorm) -------------------------------------------------------
 * y = LN(x)*gamma+beta, then optional dropout, then rows with row_zero[r]!=0 forced to 0
 * (transformers.py:115-118).  Writes y_f32 and/or y_bf16 (either may be null). */
int mmdti_layernorm_fwd(mmdti_stream_t stream, const float* x, const float* gamma, const float* beta, float eps,
                        int rows, int D, float* y_f32, void* y_bf16, float* mean, float* rstd,
                        const unsigned char* row_zero, float drop_p, unsigned long long seed, unsigned int site,
                        int y16_f16 /* != 0: y_bf16 receives fp16 (the fp16 forward-operand mode) */);
/* dx = dres + LN'(dy + dy_add) ; dgamma/dbeta atomic accumulate.  dy is fp32 (dy_dtype=MMDTI_DT_F32) or bf16;
 * dy_add (fp32, nullable) is a second upstream gradient of the LN OUTPUT (post-LN residual: a = LN(y) feeds both the FFN
 * and the next residual add); dres (fp32, nullable) is a gradient of the LN INPUT that bypasses the LN (pre-LN residual). */
/* dx_bf16 (nullable): a second copy of dx for the next backward GEMM, with the dropout-backward of site2 (probability
 * drop2_p, same seed) applied and rounded to bf16 -- saves the separate cast pass over dx.  dx_colsum (nullable, [D] fp32,
 * +=): column sums of that bf16 copy = the bias gradient of the Linear whose output gradient it is. */
int mmdti_layernorm_bwd(mmdti_stream_t stream, const void* dy, int dy_dtype, const float* dy_add, const float* x, const float* gamma,
                        const float* mean, const float* rstd, int rows, int D, const float* dres, float* dx,
                        float* dgamma, float* dbeta, const unsigned char* row_zero, float drop_p,
                        unsigned long long seed, unsigned int site, void* dx_bf16, float drop2_p, unsigned int site2,
                        float* dx_colsum);

/* ---- small utilities ------------------------------------------------------------------------ */
/* out[c] += sum_r x[r,c]  (bias gradients of every Linear) */
int mmdti_colsum_bf16(mmdti_stream_t stream, const void* x_bf16, int rows, int cols, int ld, float* out);
/* y = bf16(dropout(x)) elementwise; dropout site indexes elements 0..n-1 */
int mmdti_cast_f32_bf16(mmdti_stream_t stream, const float* x, void* y_bf16, long long n, float drop_p,
                        unsigned long long seed, unsigned int site);
int mmdti_cast_bf16_f32(mmdti_stream_t stream, const void* x_bf16, float* y, long long n);
/* fp16 forward-operand mode: y = fp16(dropout(x)) (as mmdti_cast_f32_bf16), and fp16 -> bf16 (round to nearest even) for the
 * backward GEMMs, which keep bf16 operands (gradients need bf16's range) */
int mmdti_cast_f32_f16(mmdti_stream_t stream, const float* x, void* y_f16, long long n, float drop_p, unsigned long long seed,
                       unsigned int site);
int mmdti_cast_f16_bf16(mmdti_stream_t stream, const void* x_f16, int rows, int cols, int ldx, void* y_bf16);
/* y = dropout(x) in fp32 (F.dropout on fp32 activations: infonce.py:24, mm_model.py:390-391,79,82) */
int mmdti_dropout_f32(mmdti_stream_t stream, const float* x, float* y, long long n, float drop_p,
                      unsigned long long seed, unsigned int site);
/* y += a*x (fp32) */
int mmdti_axpy_f32(mmdti_stream_t stream, const float* x, float* y, long long n, float a);

/* ---- Embedding (nn.Embedding: mm_model.py:552; HF roberta embeddings) ---------------------- */
int mmdti_embedding_fwd(mmdti_stream_t stream, const long long* ids, const float* table, long long n, int D,
                        int vocab, float* out, int accumulate);
/* dtable[ids[i]] += dout[i] (atomic), rows with ids==padding_idx skipped (padding_idx<0: none) */
int mmdti_embedding_bwd(mmdti_stream_t stream, const long long* ids, const float* dout, long long n, int D,
                        int vocab, long long padding_idx, float* dtable);
/* out[i] = table_a[ids_a[i]] + table_b[ids_b[i]] + table_c[ids_c[i]] in one pass (HF RobertaEmbeddings.forward,
 * modeling_roberta.py:75-122: word + position + token-type, summed in that order); ids_c == NULL: row 0 of table_c. */
int mmdti_embedding_fwd3(mmdti_stream_t stream, const long long* ids_a, const float* table_a, int vocab_a, const long long* ids_b,
                         const float* table_b, int vocab_b, const long long* ids_c, const float* table_c, int vocab_c, long long n, int D,
                         float* out);
/* One-hot rows for the embedding backward: out[i, ids[i]] = 1 (bf16), everything else 0; rows with ids == padding_idx or
 * out of range are all-zero.  out is [n, ld] with ld >= vocab, ld % 8 == 0.  dtable = onehot^T . dout is then ONE split-K
 * MFMA GEMM (mmdti_gemm_bf16, both operands k-major) instead of n*D contended atomics on a few hundred table rows. */
int mmdti_onehot_bf16(mmdti_stream_t stream, const long long* ids, long long n, int vocab, int ld, long long padding_idx,
                      void* out_bf16);
/* RoBERTa position ids: cumsum(ids!=pad)*(ids!=pad)+pad, int64, bit-exact (modeling_roberta.py:142-155) */
int mmdti_roberta_position_ids(mmdti_stream_t stream, const long long* ids, int B, int L, long long pad_idx,
                               long long* out);

/* ---- Gaussian pair-distance basis (GaussianLayer.forward, mm_model.py:254-269; gaussian :211-224) */
int mmdti_gbf_features_fwd(mmdti_stream_t stream, const float* dist, const long long* edge_type, const float* mul,
                           const float* bias, const float* means, const float* stds, long long P, int K, int E,
                           void* feat_bf16);
int mmdti_gbf_features_bwd(mmdti_stream_t stream, const float* dist, const long long* edge_type, const float* mul,
                           const float* bias, const float* means, const float* stds, long long P, int K, int E,
                           const void* dfeat_bf16, float* dmul, float* dbias, float* dmeans, float* dstds);
/* gbf -> gbf_proj (Linear+GELU+Linear, NonLinearHead mm_model.py:190-208) -> permute (mm_model.py:553-556) in one kernel:
 * out[b,h,i,j] (fp32, [B,H,N,ld], pad columns j >= N written 0).  w1: [F,K] bf16, w2: [H,F] bf16; built for K=128, F=128,
 * H=64.  feat/u/h (nullable, together): the [B*N*N, 128] bf16 basis / pre-activation / hidden rows for the backward.
 * tiled != 0: out is [B,H,plane] in the blocked-row tile layout the pair-attention kernels stream (see mmdti_pair_attn_fwd, layout 1)
 * and the pad keys N .. N4-1 of every query are written -inf. */
int mmdti_gbf_bias_fwd(mmdti_stream_t stream, const float* dist, const void* edge_type,
                       int edge_bytes /* 8, 4 or 2: int64 as the reference collates (mm_model.py:657-660), or narrowed */, const float* mul,
                       const float* bias, const float* means, const float* stds, const void* w1_bf16, const float* b1,
                       const void* w2_bf16, const float* b2, int B, int N, int ld, int K, int F, int H, int E, void* out,
                       void* feat_bf16, void* u_bf16, void* h_bf16, int flags /* bit 0: tiled pair layout; bit 1: u_bf16 receives
                       gelu'(pre-activation) instead of the pre-activation (then pass the same bit to mmdti_gbf_bias_bwd); bit 2 (with
                       bit 0): the pair tensor of the call is a 16-bit plane -- out is fp16 here (layout 3 of mmdti_pair_attn_fwd,
                       which explains the type); g of the two backward entry points is bf16 (layout 7 of mmdti_pair_attn_bwd).
                       Without bit 2 out and g are fp32 */,
                       const int* tile_prefix /* nullable, tiled layout only: ragged batches.  [B+1] int32 on the device,
                       tile_prefix[b+1] - tile_prefix[b] = 16-pair tiles (4x4 blocks, column block slowest) of molecule b to produce:
                       4*k_b*4*nt for its first k_b key tiles (nt = ceil(N/16); k_b as mmdti_pair_attn_fwd's key_tiles, rounded up to a
                       count its sweeps are built for).  The blocks of the all-padding key tiles behind them are not written: the ragged
                       pair-attention kernels never read them */,
                       const int* row_blocks /* nullable, with tile_prefix: PACKED token rows (mmdti_pair_attn_fwd, row_off).  [B] int32 on the
                       device: molecule b's tiles cover only its first row_blocks[b] 4-row query blocks (up to its representative pad
                       row) -- tile_prefix[b+1] - tile_prefix[b] = 4*k_b * row_blocks[b]; no query row past it is read downstream */);
/* Per-pair half of the backward of mmdti_gbf_bias_fwd, one pass over g = dL/d(out) (same layout flag): writes
 * do_bf16 [B*N*N, 64] = bf16(g re-laid out) and du_bf16 [B*N*N, 128] = bf16((do.W2) * gelu'(u)) -- the A operands of the two
 * weight-gradient GEMMs (dW2 = do^T.h, dW1 = du^T.feat; bias gradients are their column sums) -- and accumulates the
 * Gaussian-layer gradients dmul/dbias [E] and dmeans/dstds [128] (fp32, +=).  E <= 4096. */
int mmdti_gbf_bias_bwd(mmdti_stream_t stream, const void* g, const float* dist, const void* edge_type, int edge_bytes,
                       const float* mul, const float* bias, const float* means, const float* stds, const void* w1_bf16,
                       const void* w2_bf16, const void* u_bf16, int B, int N, int ld, int K, int F, int H, int E, int flags,
                       void* do_bf16, void* du_bf16, float* dmul, float* dbias, float* dmeans, float* dstds);
/* The WHOLE backward of mmdti_gbf_bias_fwd in one persistent kernel (autograd of mm_model.py:553-556 through gbf_proj
 * :190-208 and GaussianLayer :211-236): the forward saves nothing -- each 128-pair block recomputes basis / pre-activation /
 * hidden from dist and edge_type, forms do and du, and accumulates ALL parameter gradients on chip (MFMA, contraction over the
 * pairs staged transposed in LDS), flushing once per workgroup: dw1 [128,128], db1 [128], dw2 [64,128], db2 [64], dmul/dbias [E],
 * dmeans/dstds [128] (fp32, +=).  g: dL/d(out) in the layout of flags bits 0 and 2.  K=128, F=128, H=64, E <= 1536.  An fp32 g
 * enters its products as a bf16 high + a bf16 low part: the rows of g sum to zero and padded query rows repeat one basis vector, so
 * a g rounded to bf16 leaves coherent residues where the exact sums cancel. */
int mmdti_gbf_bias_bwd_full(mmdti_stream_t stream, const void* g, const float* dist, const void* edge_type, int edge_bytes,
                            const float* mul, const float* bias, const float* means, const float* stds, const void* w1_bf16,
                            const float* b1, const void* w2_bf16, int B, int N, int ld, int K, int F, int H, int E, int flags,
                            float* dw1, float* db1, float* dw2, float* db2, float* dmul, float* dbias, float* dmeans, float* dstds,
                            const int* tile_prefix /* nullable, tiled layout only: as in mmdti_gbf_bias_fwd, in this kernel's tile
                            units: tile_prefix[b+1] - tile_prefix[b] = nb * min(nb, 4*k_b) blocks of real pairs, nb = ceil(N/4) */,
                            const int* row_blocks /* nullable, with tile_prefix: packed token rows, as in the forward: the count is then
                            row_blocks[b] * min(nb, 4*k_b) -- the gradient of the query rows past the representative pad row is zero */,
                            void* workspace /* nullable; mmdti_gbf_bias_bwd_full_workspace(E) bytes, 16-byte aligned: every workgroup stores
                            its partial sums to its own slab and a second kernel folds the slabs in a fixed order -- the eight gradients
                            are then bitwise reproducible from run to run; without it the partials meet in fp32 atomics */,
                            long long workspace_bytes);
/* bytes of workspace for mmdti_gbf_bias_bwd_full with E edge types (a value, not a status code) */
int mmdti_gbf_bias_bwd_full_workspace(int E);
/* [B,N,N,H] fp32 -> [B,H,N,ld] fp32 (mm_model.py:555-556 permute(0,3,1,2).contiguous()) and its gradient
 * [B,H,N,ld] fp32 (or, tiled != 0, the blocked-row tile layout of mmdti_gbf_bias_fwd) -> [B,N,N,H] bf16 */
int mmdti_pair_permute_fwd(mmdti_stream_t stream, const float* x, float* out, int B, int N, int H, int ld);
int mmdti_pair_permute_bwd(mmdti_stream_t stream, const float* g, void* out_bf16, int B, int N, int H, int ld, int tiled);

/* ---- Pair-bias attention, head_dim 8 (unicore SelfMultiheadAttention + softmax_dropout with
 * return_attn=True, reached from transformers.py:137-139; key-padding merge :122-135) ------------
 * S = scale*q.k^T + bias_in (+ -inf at padded keys); s_out = S (next layer's bias AND the saved activation);
 * O = dropout(softmax(S)).v.   qkv: [B,N,3*H*8] bf16 (q|k|v).
 * layout (of bias_in / s_out, and of s / g in the backward):
 *   0  row-major planes [B,H,N,ld] fp32;
 *   1  tiled planes [B,H,plane] fp32, N <= 272, "blocked rows": per block of 16 queries the 4-key groups follow each other, each
 *      holding its vr query rows x 4 keys (vr = 16, or N - 16 qb in the last block) --
 *          off(query i, key j) = 16 qb N4 + vr (j - j % 4) + 4 (i % 16) + j % 4,   qb = i / 16, N4 = N rounded up to 4,
 *          plane = N * N4 rounded up to 8 elements
 *      -- so a 16x16 tile of a complete block is 256 contiguous elements in MFMA accumulator order (one contiguous KiB per wave
 *      access) and nothing is stored for queries or 4-key groups past N.  Pad keys N .. N4-1 hold -inf in S, 0 in G;
 *   3  COMPACT tiled planes: same element order, the logits chain as fp16 -- half the bytes of the forward's dominant traffic,
 *      a sixth less in the backward.  The reference carries these logits as fp16 itself when it runs under AMP (autocast makes
 *      attn_weights fp16; tasks/trainer.py:266-282).  Each layer rounds S once (to nearest even, saturating at 65504) and its own
 *      softmax runs on the rounded value, so forward and backward see the same logits.  The gradient chain g stays fp32;
 *   7  (backward only, opt-in) as 3 with g as bf16: another third less traffic, at a cost in gradient fidelity wherever sums
 *      over pairs cancel (measured: DESIGN.md); dense and ragged (key_tiles / row_off) forms like layout 3. */
int mmdti_pair_attn_fwd(mmdti_stream_t stream, const void* qkv_bf16, const void* bias_in, void* s_out,
                        void* o_bf16, const unsigned char* key_pad, int B, int N, int H, int ld, float scale,
                        float drop_p, unsigned long long seed, unsigned int site, int layout,
                        const int* key_tiles /* nullable, layout 3 only: [B] int32, number of 16-key tiles of each molecule that
                        hold a real key (ragged batches, right-padded by mm_model.py:645-682).  The key tiles past it (past the next count the
                        sweeps are built for: every count up to 9 tiles, every 2nd up to 13, every 4th beyond) are all padding:
                        they are not loaded, not computed, and not stored unless rag_store != 0 (then written as -inf: pass it for the
                        last layer, whose S goes back to the caller).  Pad QUERY rows are computed as ever. */,
                        int rag_store,
                        const int* row_off /* nullable, with key_tiles only: PACKED token rows.  [B+1] int32 on the device: molecule b owns
                        rows [row_off[b], row_off[b+1]) of qkv, o_bf16 and key_pad (then indexed by packed row) -- its n_b real tokens
                        followed by at most ONE representative pad row -- instead of rows [b*N, (b+1)*N).  Why one row is enough: the
                        reference zeroes padded rows before the first layer (transformers.py:114-118) and pads src_distance with 0 and
                        src_edge_type with the pad index (mm_model.py:657-661), so at dropout 0 every pad row of a molecule has the same
                        input, the same bias row and the same keys in every layer; the unmasked InfoNCE mean (infonce.py:32-33) then
                        weights the one row by the number of pad positions (mmdti_seq_mean_packed_fwd).  Pair planes (bias_in, s_out) stay
                        indexed by POSITION: query position i of molecule b is packed row row_off[b] + i; query rows past the
                        representative one are neither computed nor stored. */,
                        int qkv_f16 /* != 0 (layout 3 only): qkv holds fp16 and o_bf16 receives fp16 -- the fp16 forward-operand
                        mode (MMDTI_DT_AB_F16); the backward converts q | k | v on its way into LDS */);
/* g (in/out, same layout as s; fp32, or bf16 for layout 7): on entry dL/dS_l from the layers above (ignored if g_in_zero), on
 * exit dL/dS_l total = dL/d(bias_in).  dqkv: [B,N,3*H*8] bf16.  key_tiles: as in the forward; the skipped tiles of g are neither
 * read nor written (hand in a zero-initialised g for a ragged batch). */
int mmdti_pair_attn_bwd(mmdti_stream_t stream, const void* qkv_bf16, const void* s, const void* do_bf16, void* g,
                        void* dqkv_bf16, int B, int N, int H, int ld, float scale, int g_in_zero, float drop_p,
                        unsigned long long seed, unsigned int site, int layout, const int* key_tiles,
                        const int* row_off /* nullable: packed token rows of qkv / do_bf16 / dqkv_bf16, as in the forward */,
                        int qkv_f16 /* != 0 (tiled layouts): qkv holds the fp16 q | k | v the forward multiplied; the kernel rounds them to
                        bf16 (to nearest even: the values mmdti_cast_f16_bf16 would write) on their way into LDS -- the backward's
                        products keep bf16 operands, whose range activation gradients need.  do_bf16 / dqkv_bf16 stay bf16 */);

/* ---- Row softmax over materialised scores (HF eager_attention_forward :158-183; BertCoAttention
 * mm_module.py:497-514) ------------------------------------------------------------------------
 * s: [B*heads*Lq, ld] fp32 (already scaled); key_add: [B,Lk] fp32 additive mask or null.
 * p_bf16 = softmax (saved), pd_bf16 = dropout(p) (may equal p_bf16 when drop_p==0); pad columns [Lk,ld) = 0. */
int mmdti_softmax_fwd(mmdti_stream_t stream, const float* s, const float* key_add, void* p_bf16, void* pd_bf16,
                      int B, int heads, int Lq, int Lk, int ld, float drop_p, unsigned long long seed,
                      unsigned int site);
/* ds = scale * p*(dp' - sum(dp'*p)),  dp' = dropout-backward(dp) */
int mmdti_softmax_bwd(mmdti_stream_t stream, const void* p_bf16, const float* dp, void* ds_bf16, int B, int heads,
                      int Lq, int Lk, int ld, float scale, float drop_p, unsigned long long seed,
                      unsigned int site);

/* ---- Fused multi-head attention (HF eager_attention_forward modeling_roberta.py:158-183 via mm_model.py:562;
 * BertCoAttention mm_module.py:470-514): softmax(q.k^T * scale + key_add) -> dropout -> . v without materialising the
 * [B,heads,Lq,Lk] scores.  q: rows b*Lq+i, head h at columns [h*head_dim, (h+1)*head_dim), row stride ldq (elements);
 * k, v: rows b*Lk+j, stride ldk; ctx: [B*Lq, ldo] bf16; key_add: [B,Lk] fp32 additive mask or null.
 * stats: [B,heads,Lq,2] fp32 (row max of the logits in base-2 units, i.e. times log2 e; 1/row sum) saved for the backward -- opaque
 * to the caller.  head_dim 16, 32 or 64; Lq, Lk <= 256. */
int mmdti_attn_fwd(mmdti_stream_t stream, const void* q_bf16, const void* k_bf16, const void* v_bf16,
                   const float* key_add, void* ctx_bf16, float* stats, int B, int heads, int Lq, int Lk, int head_dim,
                   int ldq, int ldk, int ldo, float scale, float drop_p, unsigned long long seed, unsigned int site,
                   const int* q_off, const int* k_off, const int* k_cnt, int q_rows /* all nullable / 0 together: PACKED sequences
                   (right-padded batches, mm_model.py:645-682 / HF tokenizer padding=True).  q_off, k_off: [B+1] int32 on the device,
                   sequence b owns rows [off[b], off[b+1]) of the query-side (q, ctx) / key-side (k, v) row arrays -- its real tokens,
                   then at most one representative pad row; k_cnt: [B] int32, REAL keys of sequence b (its first k_cnt[b] key-side
                   rows).  Padded keys get probability exactly 0 in the reference (finfo.min / -10000 additive masks underflow), so
                   leaving them out of the key range is the same arithmetic; key_add must be null.  Lq, Lk: the longest sequence
                   (rows) of each side, <= 256.  stats (and drow in the backward) are then [heads, q_rows] with q_rows = q_off[B]. */,
                   int ctx_f16 /* != 0: ctx_bf16 receives fp16 (it feeds the output projection's GEMM: fp16 forward-operand mode) */);
/* dq/dk/dv (bf16, strides lddq / lddk / lddk) from dctx (stride ldo); drow: [B,heads,Lq] fp32 scratch that receives
 * sum_j dP'_ij p_ij.  Same (seed, site) as the forward call regenerates the dropout mask. */
int mmdti_attn_bwd(mmdti_stream_t stream, const void* q_bf16, const void* k_bf16, const void* v_bf16,
                   const float* key_add, const void* dctx_bf16, const float* stats, float* drow, void* dq_bf16,
                   void* dk_bf16, void* dv_bf16, int B, int heads, int Lq, int Lk, int head_dim, int ldq, int ldk, int ldo,
                   int lddq, int lddk, float scale, float drop_p, unsigned long long seed, unsigned int site,
                   const int* q_off, const int* k_off, const int* k_cnt, int q_rows /* packed sequences, as in the forward; the
                   key-side rows past k_cnt[b] (the representative pad row) receive dk = dv = 0 */);

/* ---- GELU on bf16 (kept for un-fused call sites) ------------------------------------------- */
int mmdti_gelu_fwd_bf16(mmdti_stream_t stream, const void* u_bf16, void* y_bf16, long long n);

/* ---- InfoNCE head (models/infonce.py) ------------------------------------------------------- */
/* unmasked mean over dim 1 (infonce.py:32-33): x [B,S,ld] bf16 -> out [B,D] fp32 */
int mmdti_seq_mean_fwd(mmdti_stream_t stream, const void* x_bf16, int B, int S, int D, int ld, float* out);
/* dx[b,s,:] = dout[b,:]/S  (bf16, [B,S,ld]; pad columns zeroed) */
/* dx[b*S+s, d] = bf16(dout[b, d] / S * f(aux[b*S+s, d])): f = 1 (aux_mode 0), aux (1: a saved gelu') or gelu'(aux) (2) -- the
 * backward of "pool the GELU outputs, then project" (mean_t(W2 h_t + b2) = W2 mean_t(h_t) + b2, infonce.py:28-33) */
int mmdti_seq_mean_bwd(mmdti_stream_t stream, const float* dout, int B, int S, int D, int ld, void* dx_bf16, const void* aux_bf16,
                       int ld_aux, int aux_mode);
/* The same unmasked mean over a PACKED token layout (mmdti_pair_attn_fwd, row_off): sequence b owns rows [row_off[b], row_off[b+1])
 * of x -- n_real[b] real tokens, then (if n_real[b] < S) ONE representative pad row that stands for all S - n_real[b] padded
 * positions:  out[b] = (sum_real x + (S - n_real[b]) * x_pad) / S   -- infonce.py:32-33 over the padded [B,S,*] tensor, whose pad rows
 * are identical at dropout 0.  x: [rows, ld] bf16, D % 8 == 0, ld % 8 == 0. */
int mmdti_seq_mean_packed_fwd(mmdti_stream_t stream, const void* x_bf16, int B, int S, int D, int ld, const int* row_off,
                              const int* n_real, float* out);
/* dx[r] = bf16(dout[b(r)] * w(r) / S * f(aux[r])), w = 1 for a real row, S - n_real[b] for the representative pad row; row_seq:
 * [rows] int32, the sequence each packed row belongs to; f / aux_mode as in mmdti_seq_mean_bwd. */
int mmdti_seq_mean_packed_bwd(mmdti_stream_t stream, const float* dout, int rows, int S, int D, int ld, const int* row_off,
                              const int* n_real, const int* row_seq, void* dx_bf16, const void* aux_bf16, int ld_aux, int aux_mode);
/* F.normalize(x, dim=-1) (infonce.py:104-105; contrastive.py:22-23) */
int mmdti_l2norm_fwd(mmdti_stream_t stream, const float* x, int B, int D, int ldx, float* xhat, float* inv_norm);
int mmdti_l2norm_bwd(mmdti_stream_t stream, const float* dxhat, const float* xhat, const float* inv_norm, int B,
                     int D, int ldx, float* dx);
/* One direction of the symmetric CE (infonce.py:93-98) for anchors rows [row0,row0+Bl) of qh_all against all
 * Bg keys kh_all (global negatives under DDP).  loss_sum += sum_i CE_i (the rows' terms added in row order by one thread of the
 * column pass: reproducible to the bit); dq_all rows [row0,row0+Bl) and all
 * Bg rows of dk_all are accumulated (+=, caller zero-initialises).  Gradients are of (1/(2*Bg)) * sum_i CE_i.
 * scratch: Bl*(Bg+1) floats (the logit-gradient matrix handed from the row pass to the column pass + the Bl per-row loss terms).
 * Feature width D <= 64 (the reference's 50): the Bl x Bg similarity matrix and both gradient products run as fp32 MFMAs
 * (v_mfma_f32_16x16x4_f32 -- fp32 products and accumulation, the loss keeps fp32 precision); wider D takes scalar kernels. */
int mmdti_infonce_dir(mmdti_stream_t stream, const float* qh_all, const float* kh_all, int Bg, int D, int row0,
                      int Bl, float temperature, float* loss_sum, float* dq_all, float* dk_all, float* scratch);

/* ---- ConR / SupCon (models/contrastive.py:3-59, 62-112, 114-169) --------------------------- */
/* fhat: L2-normalised features [B,D].  labels_f: [B] fp32 (regress: mean target; single: class id as float);
 * labels_i: [B,C] int64 (multi).  pred: [B] fp32 (regress).  weights: [B] fp32 or null.
 * Outputs: loss (scalar, overwritten), G [B,B] = dL/dprod. */
int mmdti_ct_loss_fwd(mmdti_stream_t stream, int mode, const float* fhat, int B, int D, const float* labels_f,
                      const long long* labels_i, int C, const float* pred, const float* weights, float w, float t,
                      float e, float coef, float* loss, float* G,
                      float* row_ws /* nullable, [B] floats: the per-row loss terms, summed in row order (bitwise reproducible loss);
                                       null: fp32 atomics */);
/* dfhat[i] = (1/t) * sum_j (G[i,j]+G[j,i]) fhat[j] */
int mmdti_ct_loss_bwd(mmdti_stream_t stream, const float* fhat, const float* G, int B, int D, float t, float* dfhat);

/* ---- FDS (models/fds.py; utils/util.py:159-169) --------------------------------------------- */
/* bins[i] = int((label_i - min_value)//bin_width) in fp32 floor-division arithmetic (fds.py:125,164), int32;
 * flags[0] = any(bin==bucket_start), flags[1] = any(bin==bucket_num-1) (flags must be zeroed by the caller). */
int mmdti_fds_bins(mmdti_stream_t stream, const float* labels, int n, float min_value, float bin_width,
                   int bucket_start, int bucket_num, int* bins, int* flags);
/* FDS.smooth (fds.py:157-190): y = calibrate_mean_var(x, m1[b], v1[b], m2[b], v2[b]) per row bucket; scale_out[r,c]
 * = d y/d x (for backward).  Stats arrays are [bucket_num-bucket_start, D]. */
int mmdti_fds_smooth(mmdti_stream_t stream, const float* x, const int* bins, const int* flags, int n, int D,
                     int bucket_start, int bucket_num, const float* m1, const float* v1, const float* m2,
                     const float* v2, float* y, float* scale_out);
/* FDS.update_running_stats (fds.py:116-155) for all buckets: per-bucket mean / unbiased var of features, EMA into
 * running_mean/var with `factor`, num_samples_tracked += count. */
int mmdti_fds_update_stats(mmdti_stream_t stream, const float* feats, const int* bins, const int* flags, int n,
                           int D, int bucket_start, int bucket_num, float factor, float* running_mean,
                           float* running_var, float* num_samples_tracked);
/* FDS._update_last_epoch_stats smoothing (fds.py:86-99): reflect-pad + conv1d across buckets */
int mmdti_fds_smooth_stats(mmdti_stream_t stream, const float* stat, int nb, int D, const float* window, int ks,
                           float* out);

/* ---- masked pooling (mm_model.py:572-576) --------------------------------------------------- */
/* pooled[b] = (sum_{valid n} a[b,n] + sum_{valid l} t[b,l]) / (n_valid_a + n_valid_t);  a:[B,Na,D], t:[B,Nt,D] fp32,
 * masks uint8 (1 = valid). */
int mmdti_masked_pool_fwd(mmdti_stream_t stream, const float* a, const float* t, const unsigned char* mask_a,
                          const unsigned char* mask_t, int B, int Na, int Nt, int D, float* pooled);
int mmdti_masked_pool_bwd(mmdti_stream_t stream, const float* dpooled, const unsigned char* mask_a,
                          const unsigned char* mask_t, int B, int Na, int Nt, int D, float* da, float* dt);

/* The same pooling over PACKED token layouts: a: [rows_a, D], t: [rows_t, D] fp32; sequence b owns rows [a_off[b], a_off[b+1]) of a,
 * the first a_cnt[b] of them real (likewise t): pooled[b] = (sum of its real rows of a and t) / (a_cnt[b] + t_cnt[b]).  The
 * representative pad rows do not enter (mm_model.py:572-573 zeroes padded rows before the sum) and get a zero gradient.
 * a_seq / t_seq: [rows] int32, the sequence of each packed row (backward only). */
int mmdti_masked_pool_packed_fwd(mmdti_stream_t stream, const float* a, const float* t, const int* a_off, const int* a_cnt,
                                 const int* t_off, const int* t_cnt, int B, int D, float* pooled);
int mmdti_masked_pool_packed_bwd(mmdti_stream_t stream, const float* dpooled, const int* a_off, const int* a_cnt, const int* a_seq,
                                 int rows_a, const int* t_off, const int* t_cnt, const int* t_seq, int rows_t, int D, float* da,
                                 float* dt);

/* ---- small fp32 linear for the classification head (mm_model.py:44-84) ---------------------- */
/* y = act(x.W^T + b); act: 0 none, 3 tanh */
int mmdti_linear_f32_fwd(mmdti_stream_t stream, const float* x, const float* W, const float* b, int rows, int in_f,
                         int out_f, int act, float* y);
/* given dy (wrt post-activation y) and y: dx (overwrite), dW/db atomic accumulate */
int mmdti_linear_f32_bwd(mmdti_stream_t stream, const float* x, const float* W, const float* y, const float* dy,
                         int rows, int in_f, int out_f, int act, float* dx, float* dW, float* db);
/* task losses (models/nnmodel.py:24-34): mean-reduced MSE / cross-entropy, loss + dlogits in one pass */
int mmdti_mse_loss(mmdti_stream_t stream, const float* pred, const float* target, int n, float* loss, float* dpred);
int mmdti_ce_loss(mmdti_stream_t stream, const float* logits, const long long* target, int B, int C, float* loss,
                  float* dlogits);
/* nn.BCEWithLogitsLoss() -- the 'bce' entry of the multilabel_classification loss table (models/nnmodel.py:28-29) -- over n = B * C
 * logits against 0 / 1 targets given as fp32: mean-reduced loss + dlogits in one pass. */
int mmdti_bce_logits_loss(mmdti_stream_t stream, const float* logits, const float* target, int n, float* loss, float* dlogits);

/* ---- optimizer step on the flat arenas (tasks/trainer.py:160,270-282) ---------------------- */
/* out[0] += sum of squares of g (zero first).  ws (nullable, ws_floats >= 1; 2048 used at most): one partial per workgroup folded in a
 * fixed order -- the gradient norm is then the same to the bit from run to run; null: fp32 atomics */
int mmdti_sumsq_f32(mmdti_stream_t stream, const float* g, long long n, float* out, float* ws, int ws_floats);
/* Adam (torch.optim.Adam semantics, eps outside sqrt of bias-corrected v): p,m,v updated in place; also refreshes the
 * bf16 shadow copy of p (the backward GEMMs' weights) and, when p_f16 is given, the fp16 one (the forward GEMMs' weights in the fp16
 * forward-operand mode; saturating).  grad is multiplied by *grad_scale_dev (device scalar, e.g. clip coefficient) if non-null.
 * step_state_dev (nullable): the device-resident schedule of mmdti_step_state_advance -- when given, lr and the bias
 * corrections are read from it and the by-value lr / step are ignored (a captured HIP graph replays with fresh values). */
int mmdti_adam_step(mmdti_stream_t stream, float* p, const float* g, float* m, float* v, void* p_bf16, long long n,
                    float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                    const float* grad_scale_dev, const float* step_state_dev, void* p_f16 /* nullable */);
/* Device-resident step state, so that a whole fine-tune step (tasks/trainer.py:177-283) can be captured in ONE HIP graph
 * and replayed: state[4] fp32 = {optimizer steps taken, this step's learning rate (HF linear warm-up / decay,
 * tasks/trainer.py:161-162), 1-beta1^t, sqrt(1-beta2^t)}; salt[2] u64 = {counter, mixed word}.  Call once at the top of
 * every step (inside the captured region): advances both. */
int mmdti_step_state_advance(mmdti_stream_t stream, float* state, unsigned long long* salt, float base_lr, int warmup_steps,
                             int total_steps, float beta1, float beta2);
/* Copies *salt into the dropout generators of every kernel library: all dropout sites launched after it on the stream XOR
 * their (seed, site) streams with it -- masks change from replay to replay although seed and site are frozen in the graph.
 * 0 (the initial value) leaves the generators exactly as the by-value seeds define them. */
int mmdti_seed_salt_pull(mmdti_stream_t stream, const unsigned long long* salt);

/* ---- hardware probes used by the test-suite ------------------------------------------------- */
/* out[64*4] <- what ds_read_b64_tr_b16 returns to each lane for an LDS image holding element index == value */
int mmdti_probe_tr_read(mmdti_stream_t stream, int row_stride_elems, unsigned short* out);

#ifdef __cplusplus
}
#endif
#endif /* MMDTI_HIP_H */
