// layers.hip -- launch sequencing of one transformer layer behind ONE call.
// At the reference's default batch sizes (16-32 molecules) a step is ~500 launches of 5-30 us kernels and the Python side of
// each (wrapper, allocations, autograd bookkeeping: 13-17 us) sets the pace, not the GPU (DESIGN.md "small batches").  The
// functions here issue the launches of a whole layer from C++: the same kernels, arguments and order as the op-by-op host path
// (functional.py), which stays as the general path -- every variant this file does not cover falls back to it -- and as the
// reference the tests hold this path bit-identical to.  No kernels of its own: it only calls the entry points of mmdti_hip.h.
#include "common.h"

using namespace mmdti;

namespace {
// dx[M, n_out] = dy[M, n_in] . w[n_in, n_out] (bf16), optionally x saved gelu' / recomputed gelu'   (ops.linear_bwd_input)
inline int dx_gemm(mmdti_stream_t s, const void* dy, int ldy, const void* w, int ldw, void* out, int M, int n_out, int n_in, int act,
                   const void* aux_in, int ld_aux) {
  return mmdti_gemm_bf16(s, dy, w, out, M, n_out, n_in, ldy, ldw, n_out, 0, 1, 1, 1, 0, 0, 0, 0, 0, 0, 1, 1.f, 0.f, nullptr, nullptr, n_out, act, aux_in,
                         nullptr, aux_in ? ld_aux : n_out, MMDTI_DT_BF16, 0.f, 0ull, 0u, nullptr, nullptr, nullptr, 0);
}
// y[M, n_out] = epi(x[M, n_in] . w[n_out, n_in]^T + bias)   (ops.linear_fwd)
inline int fwd_gemm(mmdti_stream_t s, const void* x, int ldx, const void* w, int ldw, const float* bias, void* out, int M, int n_out, int n_in,
                    int act, void* aux_out, const float* residual, int c_dtype, float drop_p, unsigned long long seed, unsigned int site) {
  return mmdti_gemm_bf16(s, x, w, out, M, n_out, n_in, ldx, ldw, n_out, 0, 0, 1, 1, 0, 0, 0, 0, 0, 0, 1, 1.f, 0.f, bias, residual, n_out, act, nullptr,
                         aux_out, n_out, c_dtype, drop_p, seed, site, nullptr, nullptr, nullptr, 0);
}
// the Linear that closes a residual branch + the LayerNorm behind it (ops.linear_ln_fwd): one kernel for 512-wide outputs up to
// ln_max_k deep, else the GEMM and the LayerNorm kernels back to back.  ln_f32 / ln_bf16: either may be null.
inline int closer(mmdti_stream_t s, const void* x, const void* w, const float* bias, const float* residual, int M, int n_out, int n_in, float drop_p,
                  unsigned long long seed, unsigned int site, float* y, const float* gamma, const float* beta, float eps, float* ln_f32,
                  void* ln_bf16, float* mean, float* rstd, int ln_max_k, int f16) {
  // f16: the fp16 forward-operand mode -- x and w hold fp16, the 16-bit LayerNorm output is fp16
  if (n_out == 512 && n_in % 64 == 0 && n_in <= ln_max_k)
    return mmdti_gemm_ln_bf16(s, x, w, bias, residual, M, n_out, n_in, n_in, n_in, n_out, drop_p, seed, site, y, gamma, beta, eps, ln_f32, ln_bf16, mean,
                              rstd, f16 ? 3 : 0);
  if (int e = fwd_gemm(s, x, n_in, w, n_in, bias, y, M, n_out, n_in, MMDTI_ACT_NONE, nullptr, residual, MMDTI_DT_F32 | (f16 ? MMDTI_DT_AB_F16 : 0), drop_p, seed,
                       site))
    return e;
  return mmdti_layernorm_fwd(s, y, gamma, beta, eps, M, n_out, ln_f32, ln_bf16, mean, rstd, nullptr, 0.f, 0ull, 0u, f16 ? 1 : 0);
}
}  // namespace

/* Forward of one Uni-Mol encoder layer behind one call: replaces the per-layer body of PairEncoderFn.forward (functional.py) with
 * the same launches -- in_proj, pair attention, out_proj + residual + dropout + LayerNorm-2, fc1 + GELU (saving gelu' or u as
 * act_fwd says), fc2 + residual + dropout and the LayerNorm that reads its output (next_mode 1: the next layer's LayerNorm-1 -> bf16;
 * 2: the encoder's final LayerNorm -> fp32; 0: none).  h1 [M,D] bf16 is this layer's LayerNorm-1 output (written by the closer of the
 * layer above or by the caller), x [M,D] fp32 the residual stream.  Every output is the caller's (they are the backward's saved
 * tensors): qkv [M,3D], s_out (pair logits, layout as s_in), o [M,D], x1 [M,D] f32, h2 [M,D], m2 / r2 [M], u / a [M,F], x_out [M,D]
 * f32, ln_out ([M,D] bf16 or f32 by next_mode), mn / rn [M]. */
extern "C" int mmdti_unimol_layer_fwd(mmdti_stream_t stream, int M, int B, int N, int H, int D, int F, int ld, float scale, float p_res,
                                      float p_att, unsigned long long seed, unsigned int site_att, unsigned int site_o, unsigned int site_f,
                                      const float* x, const void* h1, const void* s_in, const unsigned char* key_pad, int pair_layout,
                                      const int* key_tiles, int rag_store, const int* row_off, const void* w_in, const float* b_in,
                                      const void* w_out, const float* b_out, const float* g_ln2, const float* bt_ln2, float eps2,
                                      const void* w_fc1, const float* b_fc1, int act_fwd, const void* w_fc2, const float* b_fc2,
                                      int next_mode, const float* g_next, const float* bt_next, float eps_next, int ln_max_k, void* qkv,
                                      void* s_out, void* o_att, float* x1, void* h2, float* m2, float* r2, void* u_aux, void* a_act,
                                      float* x_out, void* ln_out, float* mn, float* rn, int fwd_f16) {
  MMDTI_REQUIRE(M > 0 && D > 0 && F > 0 && D % 8 == 0 && F % 8 == 0 && next_mode >= 0 && next_mode <= 2, "unimol_layer_fwd: bad shape / mode");
  MMDTI_REQUIRE(x && h1 && s_in && w_in && w_out && g_ln2 && bt_ln2 && w_fc1 && w_fc2 && qkv && s_out && o_att && x1 && h2 && m2 && r2 && u_aux && a_act && x_out,
                "unimol_layer_fwd: null argument");
  MMDTI_REQUIRE(next_mode == 0 || (g_next && bt_next && ln_out && mn && rn), "unimol_layer_fwd: the next LayerNorm needs its parameters and outputs");
  // fwd_f16 (the fp16 forward-operand mode; compact pair planes only): h1, the four weights, q | k | v, o_att, h2, a_act and a 16-bit
  // ln_out hold fp16; u_aux (the saved gelu', read by the backward) stays bf16
  MMDTI_REQUIRE(!fwd_f16 || pair_layout == 3, "unimol_layer_fwd: fp16 forward operands need the compact pair planes (layout 3)");
  const int ab = fwd_f16 ? MMDTI_DT_AB_F16 : 0, o16 = (fwd_f16 ? MMDTI_DT_F16 : MMDTI_DT_BF16) | ab;
  if (int e = fwd_gemm(stream, h1, D, w_in, D, b_in, qkv, M, 3 * D, D, MMDTI_ACT_NONE, nullptr, nullptr, o16, 0.f, 0ull, 0u)) return e;
  if (int e = mmdti_pair_attn_fwd(stream, qkv, s_in, s_out, o_att, key_pad, B, N, H, ld, scale, p_att, seed, site_att, pair_layout, key_tiles, rag_store,
                                  row_off, fwd_f16 ? 1 : 0))
    return e;
  if (int e = closer(stream, o_att, w_out, b_out, x, M, D, D, p_res, seed, site_o, x1, g_ln2, bt_ln2, eps2, nullptr, h2, m2, r2, ln_max_k, fwd_f16)) return e;
  if (int e = fwd_gemm(stream, h2, D, w_fc1, D, b_fc1, a_act, M, F, D, act_fwd, u_aux, nullptr, o16, 0.f, 0ull, 0u)) return e;
  if (next_mode == 0)
    return fwd_gemm(stream, a_act, F, w_fc2, F, b_fc2, x_out, M, D, F, MMDTI_ACT_NONE, nullptr, x1, MMDTI_DT_F32 | ab, p_res, seed, site_f);
  return closer(stream, a_act, w_fc2, b_fc2, x1, M, D, F, p_res, seed, site_f, x_out, g_next, bt_next, eps_next, next_mode == 2 ? (float*)ln_out : nullptr,
                next_mode == 1 ? ln_out : nullptr, mn, rn, ln_max_k, fwd_f16);
}

/* Backward of one Uni-Mol encoder layer (pre-LN: x1 = x + drop(out_proj(attn(LN1(x)))), x2 = x1 + drop(fc2(gelu(fc1(LN2(x1))))));
 * replaces the per-layer body of PairEncoderFn.backward (functional.py) -- transformers.py:136-139 through unicore's
 * TransformerEncoderLayer.  Eight launches: fc2 input gradient (x gelu'), fc1 input gradient, LayerNorm-2 backward, out_proj
 * input gradient, pair-attention backward, in_proj input gradient, LayerNorm-1 backward, the four weight gradients (grouped).
 *   dx_in [M,D] fp32: gradient of the layer's output; dy2 [M,D] bf16: its dropout-backward bf16 copy (written by the LayerNorm
 *   backward above).  dx_out [M,D] fp32 / dx16_out [M,D] bf16 (nullable: the lowest layer): the same two for the layer below,
 *   whose fc2 bias gradient db_below (nullable) receives the column sums of dx16_out.
 *   ws: du [M,F] | dh2 [M,D] | dy1 [M,D] | do [M,D] | dqkv [M,3D] | dh1 [M,D] (bf16) | dx_mid [M,D] fp32 | grouped-dW slabs. */
static int unimol_layer_bwd_core(mmdti_stream_t stream, int M, int B, int N, int H, int D, int F, int ld, float scale,
                                      float p_res, float p_att, unsigned long long seed, unsigned int site_f_below, unsigned int site_o,
                                      unsigned int site_att, const float* dx_in, const void* dy2, float* dx_out, void* dx16_out,
                                      float* db_below, const void* a_act, const void* u_aux, int act_dx, const void* h2, const float* x1,
                                      const float* m2, const float* r2, const void* o_att, const void* qkv, const void* s_logits,
                                      const void* h1, const float* x0, const float* m1, const float* r1, const void* w_fc2,
                                      const void* w_fc1, const void* w_out, const void* w_in, const float* g_ln2, const float* g_ln1,
                                      float* dw_fc2, float* dw_fc1, float* dw_out, float* dw_in, float* db_fc1, float* db_out,
                                      float* db_in, float* dg_ln2, float* dbt_ln2, float* dg_ln1, float* dbt_ln1, void* G,
                                      int pair_layout, int g_in_zero, const int* key_tiles, const int* row_off, void* ws,
                                      long long ws_bytes, int fwd_f16, mmdti_stream_t dw_stream, hipEvent_t ev_fork, hipEvent_t ev_done) {
  MMDTI_REQUIRE(M > 0 && D > 0 && F > 0 && D % 8 == 0 && F % 8 == 0, "unimol_layer_bwd: bad shape");
  MMDTI_REQUIRE(dx_in && dy2 && dx_out && a_act && u_aux && h2 && x1 && m2 && r2 && o_att && qkv && s_logits && h1 && x0 && m1 && r1 && w_fc2 && w_fc1 &&
                    w_out && w_in && g_ln2 && g_ln1 && dw_fc2 && dw_fc1 && dw_out && dw_in && G && ws,
                "unimol_layer_bwd: null argument");
  const long long MD = (long long)M * D, MF = (long long)M * F;
  const long long fixed = (MF + 7 * MD) * 2 + MD * 4;     // bf16 temporaries + the fp32 mid-layer gradient
  MMDTI_REQUIRE(ws_bytes >= fixed && aligned16(ws), "unimol_layer_bwd: workspace too small (%lld bytes for the temporaries alone)", fixed);
  char* wp = reinterpret_cast<char*>(ws);
  void* du = wp;                 wp += MF * 2;
  void* dh2 = wp;                wp += MD * 2;
  void* dy1 = wp;                wp += MD * 2;
  void* dob = wp;                wp += MD * 2;
  void* dqkv = wp;               wp += 3 * MD * 2;
  void* dh1 = wp;                wp += MD * 2;
  float* dx_mid = reinterpret_cast<float*>(wp); wp += MD * 4;
  void* slabs = wp;
  const long long slab_bytes = ws_bytes - fixed;
  // ---- FFN
  if (int e = dx_gemm(stream, dy2, D, w_fc2, F, du, M, F, D, act_dx, u_aux, F)) return e;
  if (int e = dx_gemm(stream, du, F, w_fc1, D, dh2, M, D, F, MMDTI_ACT_NONE, nullptr, 0)) return e;
  if (int e = mmdti_layernorm_bwd(stream, dh2, MMDTI_DT_BF16, nullptr, x1, g_ln2, m2, r2, M, D, dx_in, dx_mid, dg_ln2, dbt_ln2, nullptr, 0.f, 0ull, 0u, dy1,
                                  p_res, site_o, db_out))
    return e;
  // ---- attention
  if (int e = dx_gemm(stream, dy1, D, w_out, D, dob, M, D, D, MMDTI_ACT_NONE, nullptr, 0)) return e;
  // (fwd_f16: the saved a_act, h2, o_att, h1 and q | k | v hold fp16 -- converted inside the kernels that read them)
  if (int e = mmdti_pair_attn_bwd(stream, qkv, s_logits, dob, G, dqkv, B, N, H, ld, scale, g_in_zero, p_att, seed, site_att, pair_layout, key_tiles, row_off,
                                  fwd_f16 ? 1 : 0))
    return e;
  if (int e = dx_gemm(stream, dqkv, 3 * D, w_in, D, dh1, M, D, 3 * D, MMDTI_ACT_NONE, nullptr, 0)) return e;
  if (int e = mmdti_layernorm_bwd(stream, dh1, MMDTI_DT_BF16, nullptr, x0, g_ln1, m1, r1, M, D, dx_mid, dx_out, dg_ln1, dbt_ln1, nullptr, 0.f, 0ull, 0u,
                                  dx16_out, dx16_out ? p_res : 0.f, dx16_out ? site_f_below : 0u, dx16_out ? db_below : nullptr))
    return e;
  // ---- the four weight gradients over the same M rows: one grouped launch (bias gradients of fc1 / in_proj ride on it; those of
  //      fc2 / out_proj came from the LayerNorm backward that produced their dy)
  const void* dys[4] = {dy2, du, dy1, dqkv};
  const void* xs[4] = {a_act, h2, o_att, h1};
  float* dws[4] = {dw_fc2, dw_fc1, dw_out, dw_in};
  float* dbs[4] = {nullptr, db_fc1, nullptr, db_in};
  const int n_out[4] = {D, F, D, 3 * D}, n_in[4] = {F, D, D, D};
  const int ldy[4] = {D, F, D, 3 * D}, ldx[4] = {F, D, D, D}, lddw[4] = {F, D, D, D};
  if (!dw_stream) return mmdti_linear_dw_grouped(stream, 4, dys, xs, dws, dbs, n_out, n_in, ldy, ldx, lddw, M, slabs, slab_bytes, fwd_f16 ? 1 : 0);
  // (the stack call at small batches: the weight gradients -- leaves of the backward graph -- leave on their own stream behind
  //  ev_fork and run under the layer below; ev_done marks them finished)
  if (hipEventRecord(ev_fork, (hipStream_t)stream) != hipSuccess || hipStreamWaitEvent((hipStream_t)dw_stream, ev_fork, 0) != hipSuccess) {
    set_error("unimol_layer_bwd: event fork failed");
    return MMDTI_ERR_LAUNCH;
  }
  if (int e = mmdti_linear_dw_grouped(dw_stream, 4, dys, xs, dws, dbs, n_out, n_in, ldy, ldx, lddw, M, slabs, slab_bytes, fwd_f16 ? 1 : 0)) return e;
  if (hipEventRecord(ev_done, (hipStream_t)dw_stream) != hipSuccess) {
    set_error("unimol_layer_bwd: hipEventRecord failed");
    return MMDTI_ERR_LAUNCH;
  }
  return MMDTI_OK;
}

extern "C" int mmdti_unimol_layer_bwd(mmdti_stream_t stream, int M, int B, int N, int H, int D, int F, int ld, float scale,
                                      float p_res, float p_att, unsigned long long seed, unsigned int site_f_below, unsigned int site_o,
                                      unsigned int site_att, const float* dx_in, const void* dy2, float* dx_out, void* dx16_out,
                                      float* db_below, const void* a_act, const void* u_aux, int act_dx, const void* h2, const float* x1,
                                      const float* m2, const float* r2, const void* o_att, const void* qkv, const void* s_logits,
                                      const void* h1, const float* x0, const float* m1, const float* r1, const void* w_fc2,
                                      const void* w_fc1, const void* w_out, const void* w_in, const float* g_ln2, const float* g_ln1,
                                      float* dw_fc2, float* dw_fc1, float* dw_out, float* dw_in, float* db_fc1, float* db_out,
                                      float* db_in, float* dg_ln2, float* dbt_ln2, float* dg_ln1, float* dbt_ln1, void* G,
                                      int pair_layout, int g_in_zero, const int* key_tiles, const int* row_off, void* ws,
                                      long long ws_bytes, int fwd_f16) {
  return unimol_layer_bwd_core(stream, M, B, N, H, D, F, ld, scale, p_res, p_att, seed, site_f_below, site_o, site_att, dx_in, dy2, dx_out, dx16_out, db_below,
                               a_act, u_aux, act_dx, h2, x1, m2, r2, o_att, qkv, s_logits, h1, x0, m1, r1, w_fc2, w_fc1, w_out, w_in, g_ln2, g_ln1, dw_fc2,
                               dw_fc1, dw_out, dw_in, db_fc1, db_out, db_in, dg_ln2, dbt_ln2, dg_ln1, dbt_ln1, G, pair_layout, g_in_zero, key_tiles, row_off,
                               ws, ws_bytes, fwd_f16, nullptr, nullptr, nullptr);
}

/* The CROSS-attention variant of the same layer (BertCrossAttentionLayer, mm_module.py:615-626 through :663-677: the queries come from
 * s1, keys and values from s2): the six forward launches behind one call -- query projection, fused key | value projection, fused
 * attention (Lq queries x Lk keys per sequence; packed: q_off / k_off / k_cnt as mmdti_attn_fwd), output.dense + residual + LayerNorm,
 * intermediate + GELU, output + residual + LayerNorm.  Mq / Mk: token rows of the two sides.  Outputs as mmdti_bert_layer_fwd, with
 * q [Mq,D] and kv [Mk,2D] (bf16) in place of qkv. */
extern "C" int mmdti_bert_cross_layer_fwd(mmdti_stream_t stream, int Mq, int Mk, int B, int Lq, int Lk, int heads, int D, int F, float scale,
                                          float p_hid, float p_att, unsigned long long seed, unsigned int site_att, unsigned int site_o,
                                          unsigned int site_f, const float* s1_32, const void* s1_16, const void* s2_16, const float* key_add,
                                          const int* q_off, const int* k_off, const int* k_cnt, int q_rows, const void* w_q, const float* b_q,
                                          const void* w_kv, const float* b_kv, const void* w_o, const float* b_o, const float* g_ln1,
                                          const float* bt_ln1, const void* w_i, const float* b_i, int act_fwd, const void* w_o2,
                                          const float* b_o2, const float* g_ln2, const float* bt_ln2, float eps, int ln_max_k, void* q,
                                          void* kv, void* ctx, float* stats, float* y, float* a32, void* a16, float* am, float* ar,
                                          void* u_aux, void* i_act, float* z, float* out32, void* out16, float* zm, float* zr, int fwd_f16) {
  MMDTI_REQUIRE(Mq > 0 && Mk > 0 && D > 0 && F > 0 && heads > 0 && D % heads == 0, "bert_cross_layer_fwd: bad shape");
  MMDTI_REQUIRE(s1_32 && s1_16 && s2_16 && w_q && w_kv && w_o && g_ln1 && bt_ln1 && w_i && w_o2 && g_ln2 && bt_ln2 && q && kv && ctx && stats && y && a32 &&
                    a16 && am && ar && u_aux && i_act && z && out32 && out16 && zm && zr, "bert_cross_layer_fwd: null argument");
  const int hd = D / heads;
  const int ab = fwd_f16 ? MMDTI_DT_AB_F16 : 0;
  const char* kp = reinterpret_cast<const char*>(kv);
  if (int e = fwd_gemm(stream, s1_16, D, w_q, D, b_q, q, Mq, D, D, MMDTI_ACT_NONE, nullptr, nullptr, MMDTI_DT_BF16 | ab, 0.f, 0ull, 0u)) return e;
  if (int e = fwd_gemm(stream, s2_16, D, w_kv, D, b_kv, kv, Mk, 2 * D, D, MMDTI_ACT_NONE, nullptr, nullptr, MMDTI_DT_BF16 | ab, 0.f, 0ull, 0u)) return e;
  if (int e = mmdti_attn_fwd(stream, q, kp, kp + (size_t)D * 2, key_add, ctx, stats, B, heads, Lq, Lk, hd, D, 2 * D, D, scale, p_att, seed, site_att, q_off,
                             k_off, k_cnt, q_rows, fwd_f16 ? 1 : 0))
    return e;
  if (int e = closer(stream, ctx, w_o, b_o, s1_32, Mq, D, D, p_hid, seed, site_o, y, g_ln1, bt_ln1, eps, a32, a16, am, ar, ln_max_k, fwd_f16)) return e;
  if (int e = fwd_gemm(stream, a16, D, w_i, D, b_i, i_act, Mq, F, D, act_fwd, u_aux, nullptr, (fwd_f16 ? MMDTI_DT_F16 : MMDTI_DT_BF16) | ab, 0.f, 0ull, 0u)) return e;
  return closer(stream, i_act, w_o2, b_o2, a32, Mq, D, F, p_hid, seed, site_f, z, g_ln2, bt_ln2, eps, out32, out16, zm, zr, ln_max_k, fwd_f16);
}

/* Its backward up to the weight gradients: LayerNorm-2 backward, the FFN's two input gradients, LayerNorm-1 backward, the output
 * projection's input gradient, the fused attention backward (dq [Mq,D]; dk | dv straight into dkv [Mk,2D]), ds1 += dq . W_q (ds1 [Mq,D]
 * fp32 holds LayerNorm-1's residual gradient), ds2 = dkv . W_kv ([Mk,D] fp32, written).  The caller owns the five activation gradients
 * (dzb [Mq,D], du [Mq,F], dyb [Mq,D], dq, dkv: bf16) -- they are the A operands of the layer's weight gradients, which it launches
 * itself (two token-row counts: mmdti_linear_dw_grouped takes one per launch).  ws: da [Mq,D] | dctx [Mq,D] bf16 | dz [Mq,D] f32 | drow. */
extern "C" int mmdti_bert_cross_layer_bwd(mmdti_stream_t stream, int Mq, int Mk, int B, int Lq, int Lk, int heads, int D, int F, float scale,
                                          float p_hid, float p_att, unsigned long long seed, unsigned int site_att, unsigned int site_o,
                                          unsigned int site_f, const float* dout, float* ds1, float* ds2, const float* key_add,
                                          const int* q_off, const int* k_off, const int* k_cnt, int q_rows, const void* q, const void* kv,
                                          const float* stats, const float* y, const float* am, const float* ar, const void* u_aux,
                                          int act_dx, const float* z, const float* zm, const float* zr, const void* w_q, const void* w_kv,
                                          const void* w_o, const void* w_i, const void* w_o2, const float* g_ln1, const float* g_ln2,
                                          float* db_o, float* db_o2, float* dg_ln1, float* dbt_ln1, float* dg_ln2, float* dbt_ln2, void* dzb,
                                          void* du, void* dyb, void* dq, void* dkv, void* ws, long long ws_bytes) {
  MMDTI_REQUIRE(Mq > 0 && Mk > 0 && D > 0 && F > 0 && heads > 0 && D % heads == 0, "bert_cross_layer_bwd: bad shape");
  MMDTI_REQUIRE(dout && ds1 && ds2 && q && kv && stats && y && am && ar && u_aux && z && zm && zr && w_q && w_kv && w_o && w_i && w_o2 && g_ln1 && g_ln2 && dzb &&
                    du && dyb && dq && dkv && ws, "bert_cross_layer_bwd: null argument");
  const int hd = D / heads;
  const long long MD = (long long)Mq * D;
  const long long nrow = q_off ? (long long)heads * q_rows : (long long)B * heads * Lq;
  const long long need = 2 * MD * 2 + MD * 4 + ((nrow * 4 + 15) / 16) * 16;
  MMDTI_REQUIRE(ws_bytes >= need && aligned16(ws), "bert_cross_layer_bwd: workspace too small (%lld bytes)", need);
  char* wp = reinterpret_cast<char*>(ws);
  void* da = wp;   wp += MD * 2;
  void* dctx = wp; wp += MD * 2;
  float* dz = reinterpret_cast<float*>(wp); wp += MD * 4;
  float* drow = reinterpret_cast<float*>(wp);
  if (int e = mmdti_layernorm_bwd(stream, dout, MMDTI_DT_F32, nullptr, z, g_ln2, zm, zr, Mq, D, nullptr, dz, dg_ln2, dbt_ln2, nullptr, 0.f, 0ull, 0u, dzb, p_hid,
                                  site_f, db_o2))
    return e;
  if (int e = dx_gemm(stream, dzb, D, w_o2, F, du, Mq, F, D, act_dx, u_aux, F)) return e;
  if (int e = dx_gemm(stream, du, F, w_i, D, da, Mq, D, F, MMDTI_ACT_NONE, nullptr, 0)) return e;
  if (int e = mmdti_layernorm_bwd(stream, da, MMDTI_DT_BF16, dz, y, g_ln1, am, ar, Mq, D, nullptr, ds1, dg_ln1, dbt_ln1, nullptr, 0.f, 0ull, 0u, dyb, p_hid,
                                  site_o, db_o))
    return e;
  if (int e = dx_gemm(stream, dyb, D, w_o, D, dctx, Mq, D, D, MMDTI_ACT_NONE, nullptr, 0)) return e;
  const char* kp = reinterpret_cast<const char*>(kv);
  char* dkp = reinterpret_cast<char*>(dkv);
  if (int e = mmdti_attn_bwd(stream, q, kp, kp + (size_t)D * 2, key_add, dctx, stats, drow, dq, dkp, dkp + (size_t)D * 2, B, heads, Lq, Lk, hd, D, 2 * D, D, D,
                             2 * D, scale, p_att, seed, site_att, q_off, k_off, k_cnt, q_rows))
    return e;
  // ds1 += dq . W_q   (fp32, beta = 1);  ds2 = dkv . W_kv
  if (int e = mmdti_gemm_bf16(stream, dq, w_q, ds1, Mq, D, D, D, D, D, 0, 1, 1, 1, 0, 0, 0, 0, 0, 0, 1, 1.f, 1.f, nullptr, nullptr, D, MMDTI_ACT_NONE, nullptr,
                              nullptr, D, MMDTI_DT_F32, 0.f, 0ull, 0u, nullptr, nullptr, nullptr, 0))
    return e;
  return mmdti_gemm_bf16(stream, dkv, w_kv, ds2, Mk, D, 2 * D, 2 * D, D, D, 0, 1, 1, 1, 0, 0, 0, 0, 0, 0, 1, 1.f, 0.f, nullptr, nullptr, D, MMDTI_ACT_NONE, nullptr,
                         nullptr, D, MMDTI_DT_F32, 0.f, 0ull, 0u, nullptr, nullptr, nullptr, 0);
}

// ---------------------------------------------------------------------------------------------------------------- whole stacks
// At the reference's batch size the per-layer calls above still leave ~90 us of Python per layer and direction (17 allocations, 60
// marshalled arguments): the stack calls issue ALL layers of a tower from one call.  The saved tensors of a layer live at fixed
// offsets of one caller-owned arena (mmdti_unimol_stack_layout), the parameters come as pointer tables.
namespace {
inline long long up256(long long b) { return (b + 255) / 256 * 256; }
struct UniArena {      // byte offsets inside one layer's slice
  long long qkv, o, s, x1, h2, m2, r2, u, a, x_out, ln_out, mn, rn, stride;
  UniArena(long long M, long long D, long long F, long long s_bytes) {
    long long at = 0;
    auto take = [&](long long b) { const long long r = at; at += up256(b); return r; };
    qkv = take(M * 3 * D * 2); o = take(M * D * 2); s = take(s_bytes); x1 = take(M * D * 4); h2 = take(M * D * 2); m2 = take(M * 4); r2 = take(M * 4);
    u = take(M * F * 2); a = take(M * F * 2); x_out = take(M * D * 4); ln_out = take(M * D * 2); mn = take(M * 4); rn = take(M * 4);
    stride = at;
  }
};
// backward workspace of the stack: two layer workspaces (the side-stream weight gradients of layer l read slot l & 1 while layer
// l - 1 fills the other), a ring of three bf16 gradient copies, two fp32 gradients
struct UniBwdWs {
  long long layer_ws, lws[2], dx16[3], dx32[2], total;
  UniBwdWs(long long M, long long D, long long F, long long slab_bytes) {
    layer_ws = up256((M * F + 7 * M * D) * 2 + M * D * 4 + slab_bytes);
    long long at = 0;
    for (int i = 0; i < 2; ++i) { lws[i] = at; at += layer_ws; }
    for (int i = 0; i < 3; ++i) { dx16[i] = at; at += up256(M * D * 2); }
    for (int i = 0; i < 2; ++i) { dx32[i] = at; at += up256(M * D * 4); }
    total = at;
  }
};
}  // namespace

/* out[0] = bytes of one layer's slice of the activation arena, out[1] = bytes of the backward workspace (given the bytes of the
 * grouped weight-gradient slabs of ONE layer: mmdti_linear_dw_grouped_splits) */
extern "C" int mmdti_unimol_stack_layout(int M, int D, int F, long long s_bytes, long long dw_slab_bytes, long long* out) {
  MMDTI_REQUIRE(M > 0 && D > 0 && F > 0 && s_bytes >= 0 && dw_slab_bytes >= 0 && out, "unimol_stack_layout: bad arguments");
  out[0] = UniArena(M, D, F, s_bytes).stride;
  out[1] = UniBwdWs(M, D, F, dw_slab_bytes).total;
  return MMDTI_OK;
}

/* Forward of ALL layers of the Uni-Mol encoder (models/transformers.py:136-139 looped by :96-183) behind one call: nl x
 * mmdti_unimol_layer_fwd with the tensors a layer hands the next taken from the arena.  x0 / h1_0: the input stream and the first
 * layer's LayerNorm-1 output (the caller's).  params [nl][12]: w_in, b_in, w_out, b_out, g_ln2, bt_ln2, w_fc1, b_fc1, w_fc2, b_fc2,
 * g_ln1, bt_ln1 (16-bit forward weights).  The last layer writes the tensors the caller returns: s_last, x_last and -- with a final
 * LayerNorm (g_final non-null) -- out_final [M,D] fp32 with its statistics.  Dropout sites: site0 + 3 l + {0: attention, 1: out_proj, 2: fc2}. */
extern "C" int mmdti_unimol_stack_fwd(mmdti_stream_t stream, int nl, int M, int B, int N, int H, int D, int F, int ld, float scale, float p_res,
                                      float p_att, unsigned long long seed, unsigned int site0, const float* x0, const void* h1_0,
                                      const void* s_in, const unsigned char* key_pad, int pair_layout, const int* key_tiles,
                                      int rag_store_last, const int* row_off, const void* const* params, int act_fwd, float eps_ln,
                                      const float* g_final, const float* bt_final, float eps_final, int ln_max_k, void* arena,
                                      long long arena_bytes, long long s_bytes, float* x_last, void* s_last, float* out_final,
                                      float* mean_final, float* rstd_final, int fwd_f16) {
  MMDTI_REQUIRE(nl > 0 && params && arena && aligned16(arena) && x0 && h1_0 && s_in && x_last && s_last, "unimol_stack_fwd: null argument");
  MMDTI_REQUIRE(!g_final || (bt_final && out_final && mean_final && rstd_final), "unimol_stack_fwd: the final LayerNorm needs its outputs");
  const UniArena A(M, D, F, s_bytes);
  MMDTI_REQUIRE(arena_bytes >= A.stride * nl, "unimol_stack_fwd: arena too small (%lld bytes per layer)", A.stride);
  char* base = reinterpret_cast<char*>(arena);
  for (int l = 0; l < nl; ++l) {
    char* a = base + A.stride * l;
    const char* prev = a - A.stride;
    const void* const* P = params + 12 * l;
    const bool last = l == nl - 1;
    const float* x = l ? reinterpret_cast<const float*>(prev + A.x_out) : x0;
    const void* h1 = l ? static_cast<const void*>(prev + A.ln_out) : h1_0;
    const void* sp = l ? static_cast<const void*>(prev + A.s) : s_in;
    const int next_mode = last ? (g_final ? 2 : 0) : 1;
    const float* gn = last ? g_final : reinterpret_cast<const float*>(params[12 * (l + 1) + 10]);
    const float* bn = last ? bt_final : reinterpret_cast<const float*>(params[12 * (l + 1) + 11]);
    if (int e = mmdti_unimol_layer_fwd(
            stream, M, B, N, H, D, F, ld, scale, p_res, p_att, seed, site0 + 3 * l, site0 + 3 * l + 1, site0 + 3 * l + 2, x, h1, sp, l ? nullptr : key_pad,
            pair_layout, key_tiles, last ? rag_store_last : 0, row_off, P[0], (const float*)P[1], P[2], (const float*)P[3], (const float*)P[4],
            (const float*)P[5], eps_ln, P[6], (const float*)P[7], act_fwd, P[8], (const float*)P[9], next_mode, gn, bn, last ? eps_final : eps_ln,
            ln_max_k, a + A.qkv, last ? s_last : static_cast<void*>(a + A.s), a + A.o, reinterpret_cast<float*>(a + A.x1), a + A.h2,
            reinterpret_cast<float*>(a + A.m2), reinterpret_cast<float*>(a + A.r2), a + A.u, a + A.a,
            last ? x_last : reinterpret_cast<float*>(a + A.x_out), last ? static_cast<void*>(out_final) : static_cast<void*>(a + A.ln_out),
            last ? mean_final : reinterpret_cast<float*>(a + A.mn), last ? rstd_final : reinterpret_cast<float*>(a + A.rn), fwd_f16))
      return e;
  }
  return MMDTI_OK;
}

/* Backward of the same stack, top layer first: nl x mmdti_unimol_layer_bwd.  dx_in [M,D] fp32 / dy2_in [M,D] bf16: the gradient of
 * the top layer's output and its dropout-backward bf16 copy (from the final LayerNorm's backward); dx_final [M,D] fp32: the gradient
 * of x0.  m1_0 / r1_0: LayerNorm-1 statistics of the first layer (the caller's, like x0 / h1_0); s_last: the top layer's logits.
 * bparams [nl][6]: w_fc2, w_fc1, w_out, w_in (bf16), g_ln2, g_ln1;  grads [nl][12]: dw_fc2, dw_fc1, dw_out, dw_in, db_fc2, db_fc1,
 * db_out, db_in, dg_ln2, dbt_ln2, dg_ln1, dbt_ln1 (fp32, accumulated).  G: the pair-gradient chain (g_first_zero: not yet written).
 * dw_stream + events (hipEvent_t [3], nullable together): the weight gradients run on dw_stream under the layer below; `stream`
 * has joined dw_stream when the call returns. */
extern "C" int mmdti_unimol_stack_bwd(mmdti_stream_t stream, int nl, int M, int B, int N, int H, int D, int F, int ld, float scale, float p_res,
                                      float p_att, unsigned long long seed, unsigned int site0, const float* dx_in, const void* dy2_in,
                                      float* dx_final, const float* x0, const void* h1_0, const float* m1_0, const float* r1_0,
                                      const void* s_last, const void* const* bparams, int act_dx, void* const* grads, void* G, int pair_layout,
                                      int g_first_zero, const int* key_tiles, const int* row_off, const void* arena, long long arena_bytes,
                                      long long s_bytes, void* ws, long long ws_bytes, long long dw_slab_bytes, int fwd_f16,
                                      mmdti_stream_t dw_stream, void* const* events) {
  MMDTI_REQUIRE(nl > 0 && bparams && grads && arena && ws && aligned16(ws) && dx_in && dy2_in && dx_final && x0 && h1_0 && m1_0 && r1_0 && s_last && G,
                "unimol_stack_bwd: null argument");
  MMDTI_REQUIRE(!dw_stream || (events && events[0] && events[1] && events[2]), "unimol_stack_bwd: a weight-gradient stream needs three events");
  const UniArena A(M, D, F, s_bytes);
  const UniBwdWs W(M, D, F, dw_slab_bytes);
  MMDTI_REQUIRE(arena_bytes >= A.stride * nl && ws_bytes >= W.total, "unimol_stack_bwd: arena / workspace too small (%lld / %lld bytes)", A.stride * nl, W.total);
  const char* base = reinterpret_cast<const char*>(arena);
  char* wb = reinterpret_cast<char*>(ws);
  hipEvent_t fork = dw_stream ? (hipEvent_t)events[0] : nullptr;
  const float* dx = dx_in;
  const void* dy2 = dy2_in;
  for (int l = nl - 1, it = 0; l >= 0; --l, ++it) {
    const char* a = base + A.stride * l;
    const char* prev = a - A.stride;
    hipEvent_t done = dw_stream ? (hipEvent_t)events[1 + (it & 1)] : nullptr;
    // (the weight gradients issued two layers ago read this layer workspace and the ring slot about to be written)
    if (dw_stream && it >= 2 && hipStreamWaitEvent((hipStream_t)stream, done, 0) != hipSuccess) {
      set_error("unimol_stack_bwd: hipStreamWaitEvent failed");
      return MMDTI_ERR_LAUNCH;
    }
    float* dx_out = l ? reinterpret_cast<float*>(wb + W.dx32[it & 1]) : dx_final;
    void* dx16_out = l ? static_cast<void*>(wb + W.dx16[it % 3]) : nullptr;
    const void* const* P = bparams + 6 * l;
    void* const* Gr = grads + 12 * l;
    if (int e = unimol_layer_bwd_core(
            stream, M, B, N, H, D, F, ld, scale, p_res, p_att, seed, l ? site0 + 3 * (l - 1) + 2 : 0u, site0 + 3 * l + 1, site0 + 3 * l, dx, dy2, dx_out,
            dx16_out, l ? reinterpret_cast<float*>(grads[12 * (l - 1) + 4]) : nullptr, a + A.a, a + A.u, act_dx, a + A.h2,
            reinterpret_cast<const float*>(a + A.x1), reinterpret_cast<const float*>(a + A.m2), reinterpret_cast<const float*>(a + A.r2), a + A.o, a + A.qkv,
            l == nl - 1 ? s_last : static_cast<const void*>(a + A.s), l ? static_cast<const void*>(prev + A.ln_out) : h1_0,
            l ? reinterpret_cast<const float*>(prev + A.x_out) : x0, l ? reinterpret_cast<const float*>(prev + A.mn) : m1_0,
            l ? reinterpret_cast<const float*>(prev + A.rn) : r1_0, P[0], P[1], P[2], P[3], (const float*)P[4], (const float*)P[5], (float*)Gr[0], (float*)Gr[1],
            (float*)Gr[2], (float*)Gr[3], (float*)Gr[5], (float*)Gr[6], (float*)Gr[7], (float*)Gr[8], (float*)Gr[9], (float*)Gr[10], (float*)Gr[11], G,
            pair_layout, it == 0 ? g_first_zero : 0, key_tiles, row_off, wb + W.lws[it & 1], W.layer_ws, fwd_f16, dw_stream, fork, done))
      return e;
    dx = dx_out;
    dy2 = dx16_out;
  }
  if (dw_stream) {
    for (int i = 0; i < 2 && i < nl; ++i)
      if (hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)events[1 + i], 0) != hipSuccess) {
        set_error("unimol_stack_bwd: join failed");
        return MMDTI_ERR_LAUNCH;
      }
  }
  return MMDTI_OK;
}

/* Forward of one post-LN BERT layer with SELF-attention, the q | k | v projection as one GEMM and the fused attention kernels
 * (HF RobertaLayer reached from models/mm_model.py:562; the variant functional._bert_layer_fwd takes on the hot path): the same six
 * launches behind one call.  s1_32 / s1_16 [Mq,D]: the layer input (fp32 residual stream and its bf16 copy); w_qkv [3D,D], b_qkv [3D]
 * the fused projection.  Outputs (the backward's saved tensors): qkv [Mq,3D], ctx [Mq,D], stats, y [Mq,D] f32 (pre-LN1), a32 / a16
 * (LN1 output), am / ar, u / i_act [Mq,F], z [Mq,D] f32 (pre-LN2), out32 / out16 (LN2 output), zm / zr.  Packed sequences: q_off ... as
 * mmdti_attn_fwd (null / 0: dense, key_add [B,Lq] nullable). */
extern "C" int mmdti_bert_layer_fwd(mmdti_stream_t stream, int Mq, int B, int L, int heads, int D, int F, float scale, float p_hid, float p_att,
                                    unsigned long long seed, unsigned int site_att, unsigned int site_o, unsigned int site_f,
                                    const float* s1_32, const void* s1_16, const float* key_add, const int* q_off, const int* k_off,
                                    const int* k_cnt, int q_rows, const void* w_qkv, const float* b_qkv, const void* w_o, const float* b_o,
                                    const float* g_ln1, const float* bt_ln1, const void* w_i, const float* b_i, int act_fwd,
                                    const void* w_o2, const float* b_o2, const float* g_ln2, const float* bt_ln2, float eps, int ln_max_k,
                                    void* qkv, void* ctx, float* stats, float* y, float* a32, void* a16, float* am, float* ar, void* u_aux,
                                    void* i_act, float* z, float* out32, void* out16, float* zm, float* zr, int fwd_f16) {
  MMDTI_REQUIRE(Mq > 0 && D > 0 && F > 0 && heads > 0 && D % heads == 0, "bert_layer_fwd: bad shape");
  MMDTI_REQUIRE(s1_32 && s1_16 && w_qkv && w_o && g_ln1 && bt_ln1 && w_i && w_o2 && g_ln2 && bt_ln2 && qkv && ctx && stats && y && a32 && a16 && am && ar &&
                    u_aux && i_act && z && out32 && out16 && zm && zr, "bert_layer_fwd: null argument");
  const int hd = D / heads;
  const char* qp = reinterpret_cast<const char*>(qkv);
  // fwd_f16 (the fp16 forward-operand mode): s1_16, the four weights, ctx, a16, i_act and out16 hold fp16; q | k | v stay bf16 (the
  // attention kernels' operand type in every mode), u_aux too
  const int ab = fwd_f16 ? MMDTI_DT_AB_F16 : 0;
  if (int e = fwd_gemm(stream, s1_16, D, w_qkv, D, b_qkv, qkv, Mq, 3 * D, D, MMDTI_ACT_NONE, nullptr, nullptr, MMDTI_DT_BF16 | ab, 0.f, 0ull, 0u)) return e;
  if (int e = mmdti_attn_fwd(stream, qp, qp + (size_t)D * 2, qp + (size_t)2 * D * 2, key_add, ctx, stats, B, heads, L, L, hd, 3 * D, 3 * D, D,
                             scale, p_att, seed, site_att, q_off, k_off, k_cnt, q_rows, fwd_f16 ? 1 : 0))
    return e;
  if (int e = closer(stream, ctx, w_o, b_o, s1_32, Mq, D, D, p_hid, seed, site_o, y, g_ln1, bt_ln1, eps, a32, a16, am, ar, ln_max_k, fwd_f16)) return e;
  if (int e = fwd_gemm(stream, a16, D, w_i, D, b_i, i_act, Mq, F, D, act_fwd, u_aux, nullptr, (fwd_f16 ? MMDTI_DT_F16 : MMDTI_DT_BF16) | ab, 0.f, 0ull, 0u)) return e;
  return closer(stream, i_act, w_o2, b_o2, a32, Mq, D, F, p_hid, seed, site_f, z, g_ln2, bt_ln2, eps, out32, out16, zm, zr, ln_max_k, fwd_f16);
}

/* Backward of the same layer: LayerNorm-2 backward, the FFN's two input gradients, LayerNorm-1 backward (its output fed the FFN and
 * the residual: dy_add), the attention output projection's input gradient, the fused attention backward (dq | dk | dv written
 * straight into one [Mq,3D] gradient), the fused projection's input gradient accumulated into ds1, and the four weight gradients as
 * one grouped launch.  dout [Mq,D] fp32 -> ds1 [Mq,D] fp32 (written).
 *   ws: dzb [Mq,D] | du [Mq,F] | da [Mq,D] | dyb [Mq,D] | dctx [Mq,D] | dqkv [Mq,3D] (bf16) | dz [Mq,D] f32 | drow [heads*rows] f32 | slabs. */
extern "C" int mmdti_bert_layer_bwd(mmdti_stream_t stream, int Mq, int B, int L, int heads, int D, int F, float scale, float p_hid, float p_att,
                                    unsigned long long seed, unsigned int site_att, unsigned int site_o, unsigned int site_f,
                                    const float* dout, float* ds1, const void* s1_16, const float* key_add, const int* q_off,
                                    const int* k_off, const int* k_cnt, int q_rows, const void* qkv, const void* ctx, const float* stats,
                                    const float* y, const void* a16, const float* am, const float* ar, const void* u_aux, int act_dx,
                                    const void* i_act, const float* z, const float* zm, const float* zr, const void* w_qkv,
                                    const void* w_o, const void* w_i, const void* w_o2, const float* g_ln1, const float* g_ln2,
                                    float* dw_qkv, int lddw_qkv, float* db_qkv, float* dw_o, float* db_o, float* dw_i, float* db_i,
                                    float* dw_o2, float* db_o2, float* dg_ln1, float* dbt_ln1, float* dg_ln2, float* dbt_ln2, void* ws,
                                    long long ws_bytes, int fwd_f16) {
  MMDTI_REQUIRE(Mq > 0 && D > 0 && F > 0 && heads > 0 && D % heads == 0, "bert_layer_bwd: bad shape");
  MMDTI_REQUIRE(dout && ds1 && s1_16 && qkv && ctx && stats && y && a16 && am && ar && u_aux && i_act && z && zm && zr && w_qkv && w_o && w_i && w_o2 &&
                    g_ln1 && g_ln2 && dw_qkv && dw_o && dw_i && dw_o2 && ws, "bert_layer_bwd: null argument");
  const int hd = D / heads;
  const long long MD = (long long)Mq * D, MF = (long long)Mq * F;
  const long long nrow = q_off ? (long long)heads * q_rows : (long long)B * heads * L;
  const long long fixed = (MF + 7 * MD) * 2 + MD * 4 + ((nrow * 4 + 15) / 16) * 16;
  MMDTI_REQUIRE(ws_bytes >= fixed && aligned16(ws), "bert_layer_bwd: workspace too small (%lld bytes for the temporaries alone)", fixed);
  char* wp = reinterpret_cast<char*>(ws);
  void* dzb = wp;  wp += MD * 2;
  void* du = wp;   wp += MF * 2;
  void* da = wp;   wp += MD * 2;
  void* dyb = wp;  wp += MD * 2;
  void* dctx = wp; wp += MD * 2;
  char* dqkv = wp; wp += 3 * MD * 2;
  float* dz = reinterpret_cast<float*>(wp); wp += MD * 4;
  float* drow = reinterpret_cast<float*>(wp); wp += ((nrow * 4 + 15) / 16) * 16;
  void* slabs = wp;
  const long long slab_bytes = ws_bytes - fixed;
  if (int e = mmdti_layernorm_bwd(stream, dout, MMDTI_DT_F32, nullptr, z, g_ln2, zm, zr, Mq, D, nullptr, dz, dg_ln2, dbt_ln2, nullptr, 0.f, 0ull, 0u, dzb, p_hid,
                                  site_f, db_o2))
    return e;
  if (int e = dx_gemm(stream, dzb, D, w_o2, F, du, Mq, F, D, act_dx, u_aux, F)) return e;
  if (int e = dx_gemm(stream, du, F, w_i, D, da, Mq, D, F, MMDTI_ACT_NONE, nullptr, 0)) return e;
  if (int e = mmdti_layernorm_bwd(stream, da, MMDTI_DT_BF16, dz, y, g_ln1, am, ar, Mq, D, nullptr, ds1, dg_ln1, dbt_ln1, nullptr, 0.f, 0ull, 0u, dyb, p_hid,
                                  site_o, db_o))
    return e;
  if (int e = dx_gemm(stream, dyb, D, w_o, D, dctx, Mq, D, D, MMDTI_ACT_NONE, nullptr, 0)) return e;
  const char* qp = reinterpret_cast<const char*>(qkv);
  if (int e = mmdti_attn_bwd(stream, qp, qp + (size_t)D * 2, qp + (size_t)2 * D * 2, key_add, dctx, stats, drow, dqkv, dqkv + (size_t)D * 2,
                             dqkv + (size_t)2 * D * 2, B, heads, L, L, hd, 3 * D, 3 * D, D, 3 * D, 3 * D, scale, p_att, seed, site_att, q_off,
                             k_off, k_cnt, q_rows))
    return e;
  // ds1 += dqkv . W_qkv   (fp32, beta = 1)
  if (int e = mmdti_gemm_bf16(stream, dqkv, w_qkv, ds1, Mq, D, 3 * D, 3 * D, D, D, 0, 1, 1, 1, 0, 0, 0, 0, 0, 0, 1, 1.f, 1.f, nullptr, nullptr, D, MMDTI_ACT_NONE,
                              nullptr, nullptr, D, MMDTI_DT_F32, 0.f, 0ull, 0u, nullptr, nullptr, nullptr, 0))
    return e;
  const void* dys[4] = {dqkv, dzb, du, dyb};
  const void* xs[4] = {s1_16, i_act, a16, ctx};
  float* dws[4] = {dw_qkv, dw_o2, dw_i, dw_o};
  float* dbs[4] = {db_qkv, nullptr, db_i, nullptr};
  const int n_out[4] = {3 * D, D, F, D}, n_in[4] = {D, F, D, D};
  const int ldy[4] = {3 * D, D, F, D}, ldx[4] = {D, F, D, D}, lddw[4] = {lddw_qkv, F, D, D};
  // (fwd_f16: the saved s1_16, i_act, a16 and ctx hold fp16 -- converted between LDS and the matrix pipe)
  return mmdti_linear_dw_grouped(stream, 4, dys, xs, dws, dbs, n_out, n_in, ldy, ldx, lddw, Mq, slabs, slab_bytes, fwd_f16 ? 1 : 0);
}

// ---------------------------------------------------------------------------------------------------------------- tower 2's stack
namespace {
struct BertArena {     // byte offsets inside one layer's slice
  long long qkv, ctx, stats, y, a32, a16, am, ar, u, i, z, out32, out16, zm, zr, stride;
  BertArena(long long M, long long D, long long F, long long stats_bytes) {
    long long at = 0;
    auto take = [&](long long b) { const long long r = at; at += up256(b); return r; };
    qkv = take(M * 3 * D * 2); ctx = take(M * D * 2); stats = take(stats_bytes); y = take(M * D * 4); a32 = take(M * D * 4); a16 = take(M * D * 2);
    am = take(M * 4); ar = take(M * 4); u = take(M * F * 2); i = take(M * F * 2); z = take(M * D * 4); out32 = take(M * D * 4); out16 = take(M * D * 2);
    zm = take(M * 4); zr = take(M * 4);
    stride = at;
  }
};
struct BertBwdWs {     // one layer workspace (mmdti_bert_layer_bwd's) + two fp32 gradients handed from layer to layer
  long long layer_ws, lws, ds[2], total;
  BertBwdWs(long long M, long long D, long long F, long long nrow, long long slab_bytes) {
    layer_ws = up256((M * F + 7 * M * D) * 2 + M * D * 4 + ((nrow * 4 + 15) / 16) * 16 + slab_bytes);
    lws = 0;
    long long at = layer_ws;
    for (int k = 0; k < 2; ++k) { ds[k] = at; at += up256(M * D * 4); }
    total = at;
  }
};
}  // namespace

/* out[0] = arena bytes per layer, out[1] = backward workspace bytes (stats_bytes: one layer's softmax statistics; nrow: rows of the
 * attention backward's row term -- heads * q_rows packed, B * heads * L dense; dw_slab_bytes: see mmdti_unimol_stack_layout) */
extern "C" int mmdti_bert_stack_layout(int Mq, int D, int F, long long stats_bytes, long long nrow, long long dw_slab_bytes, long long* out) {
  MMDTI_REQUIRE(Mq > 0 && D > 0 && F > 0 && stats_bytes >= 0 && nrow >= 0 && dw_slab_bytes >= 0 && out, "bert_stack_layout: bad arguments");
  out[0] = BertArena(Mq, D, F, stats_bytes).stride;
  out[1] = BertBwdWs(Mq, D, F, nrow, dw_slab_bytes).total;
  return MMDTI_OK;
}

/* Forward of ALL layers of tower 2 (HF RobertaEncoder's layer loop, reached from models/mm_model.py:562) behind one call: nl x
 * mmdti_bert_layer_fwd, a layer's out32 / out16 being the next one's input.  s1_32_0 / s1_16_0: the embeddings' LayerNorm output (the
 * caller's).  params [nl][12]: w_qkv, b_qkv, w_o, b_o, g_ln1, bt_ln1, w_i, b_i, w_o2, b_o2, g_ln2, bt_ln2 (16-bit forward weights,
 * q | k | v fused).  The last layer's fp32 output goes to out32_last (the caller's).  Dropout sites: site0 + 3 l + {0: attention, 1, 2}. */
extern "C" int mmdti_bert_stack_fwd(mmdti_stream_t stream, int nl, int Mq, int B, int L, int heads, int D, int F, float scale, float p_hid,
                                    float p_att, unsigned long long seed, unsigned int site0, const float* s1_32_0, const void* s1_16_0,
                                    const float* key_add, const int* q_off, const int* k_off, const int* k_cnt, int q_rows,
                                    const void* const* params, int act_fwd, float eps, int ln_max_k, void* arena, long long arena_bytes,
                                    long long stats_bytes, float* out32_last, int fwd_f16) {
  MMDTI_REQUIRE(nl > 0 && params && arena && aligned16(arena) && s1_32_0 && s1_16_0 && out32_last, "bert_stack_fwd: null argument");
  const BertArena A(Mq, D, F, stats_bytes);
  MMDTI_REQUIRE(arena_bytes >= A.stride * nl, "bert_stack_fwd: arena too small (%lld bytes per layer)", A.stride);
  char* base = reinterpret_cast<char*>(arena);
  for (int l = 0; l < nl; ++l) {
    char* a = base + A.stride * l;
    const char* prev = a - A.stride;
    const void* const* P = params + 12 * l;
    const bool last = l == nl - 1;
    if (int e = mmdti_bert_layer_fwd(
            stream, Mq, B, L, heads, D, F, scale, p_hid, p_att, seed, site0 + 3 * l, site0 + 3 * l + 1, site0 + 3 * l + 2,
            l ? reinterpret_cast<const float*>(prev + A.out32) : s1_32_0, l ? static_cast<const void*>(prev + A.out16) : s1_16_0, key_add, q_off, k_off, k_cnt,
            q_rows, P[0], (const float*)P[1], P[2], (const float*)P[3], (const float*)P[4], (const float*)P[5], P[6], (const float*)P[7], act_fwd, P[8],
            (const float*)P[9], (const float*)P[10], (const float*)P[11], eps, ln_max_k, a + A.qkv, a + A.ctx, reinterpret_cast<float*>(a + A.stats),
            reinterpret_cast<float*>(a + A.y), reinterpret_cast<float*>(a + A.a32), a + A.a16, reinterpret_cast<float*>(a + A.am),
            reinterpret_cast<float*>(a + A.ar), a + A.u, a + A.i, reinterpret_cast<float*>(a + A.z), last ? out32_last : reinterpret_cast<float*>(a + A.out32),
            a + A.out16, reinterpret_cast<float*>(a + A.zm), reinterpret_cast<float*>(a + A.zr), fwd_f16))
      return e;
  }
  return MMDTI_OK;
}

/* Backward of the same stack, top layer first: nl x mmdti_bert_layer_bwd.  dout [Mq,D] fp32: the gradient of the tower's output;
 * ds1_final [Mq,D] fp32: the gradient of s1_32_0.  bparams [nl][6]: w_qkv, w_o, w_i, w_o2 (bf16), g_ln1, g_ln2;  grads [nl][12]:
 * dw_qkv, db_qkv, dw_o, db_o, dw_i, db_i, dw_o2, db_o2, dg_ln1, dbt_ln1, dg_ln2, dbt_ln2 (fp32, accumulated; lddw_qkv: row stride of
 * the fused q | k | v weight gradient). */
extern "C" int mmdti_bert_stack_bwd(mmdti_stream_t stream, int nl, int Mq, int B, int L, int heads, int D, int F, float scale, float p_hid,
                                    float p_att, unsigned long long seed, unsigned int site0, const float* dout, float* ds1_final,
                                    const void* s1_16_0, const float* key_add, const int* q_off, const int* k_off, const int* k_cnt,
                                    int q_rows, const void* const* bparams, int act_dx, void* const* grads, int lddw_qkv, const void* arena,
                                    long long arena_bytes, long long stats_bytes, void* ws, long long ws_bytes, long long dw_slab_bytes,
                                    int fwd_f16) {
  MMDTI_REQUIRE(nl > 0 && bparams && grads && arena && ws && aligned16(ws) && dout && ds1_final && s1_16_0, "bert_stack_bwd: null argument");
  const BertArena A(Mq, D, F, stats_bytes);
  const long long nrow = q_off ? (long long)heads * q_rows : (long long)B * heads * L;
  const BertBwdWs W(Mq, D, F, nrow, dw_slab_bytes);
  MMDTI_REQUIRE(arena_bytes >= A.stride * nl && ws_bytes >= W.total, "bert_stack_bwd: arena / workspace too small (%lld / %lld bytes)", A.stride * nl, W.total);
  const char* base = reinterpret_cast<const char*>(arena);
  char* wb = reinterpret_cast<char*>(ws);
  const float* d = dout;
  for (int l = nl - 1, it = 0; l >= 0; --l, ++it) {
    const char* a = base + A.stride * l;
    const char* prev = a - A.stride;
    float* ds1 = l ? reinterpret_cast<float*>(wb + W.ds[it & 1]) : ds1_final;
    const void* const* P = bparams + 6 * l;
    void* const* G = grads + 12 * l;
    if (int e = mmdti_bert_layer_bwd(
            stream, Mq, B, L, heads, D, F, scale, p_hid, p_att, seed, site0 + 3 * l, site0 + 3 * l + 1, site0 + 3 * l + 2, d, ds1,
            l ? static_cast<const void*>(prev + A.out16) : s1_16_0, key_add, q_off, k_off, k_cnt, q_rows, a + A.qkv, a + A.ctx,
            reinterpret_cast<const float*>(a + A.stats), reinterpret_cast<const float*>(a + A.y), a + A.a16, reinterpret_cast<const float*>(a + A.am),
            reinterpret_cast<const float*>(a + A.ar), a + A.u, act_dx, a + A.i, reinterpret_cast<const float*>(a + A.z),
            reinterpret_cast<const float*>(a + A.zm), reinterpret_cast<const float*>(a + A.zr), P[0], P[1], P[2], P[3], (const float*)P[4], (const float*)P[5],
            (float*)G[0], lddw_qkv, (float*)G[1], (float*)G[2], (float*)G[3], (float*)G[4], (float*)G[5], (float*)G[6], (float*)G[7], (float*)G[8],
            (float*)G[9], (float*)G[10], (float*)G[11], wb + W.lws, W.layer_ws, fwd_f16))
      return e;
    d = ds1;
  }
  return MMDTI_OK;
}
