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
}  // namespace

/* Backward of one Uni-Mol encoder layer (pre-LN: x1 = x + drop(out_proj(attn(LN1(x)))), x2 = x1 + drop(fc2(gelu(fc1(LN2(x1))))));
 * replaces the per-layer body of PairEncoderFn.backward (functional.py) -- transformers.py:136-139 through unicore's
 * TransformerEncoderLayer.  Eight launches: fc2 input gradient (x gelu'), fc1 input gradient, LayerNorm-2 backward, out_proj
 * input gradient, pair-attention backward, in_proj input gradient, LayerNorm-1 backward, the four weight gradients (grouped).
 *   dx_in [M,D] fp32: gradient of the layer's output; dy2 [M,D] bf16: its dropout-backward bf16 copy (written by the LayerNorm
 *   backward above).  dx_out [M,D] fp32 / dx16_out [M,D] bf16 (nullable: the lowest layer): the same two for the layer below,
 *   whose fc2 bias gradient db_below (nullable) receives the column sums of dx16_out.
 *   ws: du [M,F] | dh2 [M,D] | dy1 [M,D] | do [M,D] | dqkv [M,3D] | dh1 [M,D] (bf16) | dx_mid [M,D] fp32 | grouped-dW slabs. */
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
                                      long long ws_bytes) {
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
  if (int e = mmdti_pair_attn_bwd(stream, qkv, s_logits, dob, G, dqkv, B, N, H, ld, scale, g_in_zero, p_att, seed, site_att, pair_layout, key_tiles, row_off))
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
  return mmdti_linear_dw_grouped(stream, 4, dys, xs, dws, dbs, n_out, n_in, ldy, ldx, lddw, M, slabs, slab_bytes);
}
