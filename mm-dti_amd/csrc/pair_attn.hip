// Pair-bias multi-head attention for the Uni-Mol tower (64 heads x head_dim 8), forward and backward.
//
// Replaces unicore SelfMultiheadAttention + softmax_dropout (fused CUDA op upstream) as reached from
// models/transformers.py:137-139 with return_attn=True, including the key-padding merge of :122-135:
//     S_l = scale * q.k^T + S_{l-1}          (S_0 = Gaussian pair bias, -inf at padded keys)
//     O   = dropout(softmax(S_l)) . v
// S_l is an OUTPUT (it is the next layer's bias and the activation saved for backward), so the score tile is
// never kept on chip only: the kernel is a stream over the [B,H,N,N] pair tensor and is HBM-bound
// (head_dim 8 => 32 flop per 8 B of pair traffic).  One workgroup per (molecule, head); lanes own KEYS
// (K/V rows live in registers for the whole tile), waves walk query rows, so every pair-tensor access is a
// fully coalesced row segment and dK/dV need no cross-lane traffic.
//
// Backward recomputes P from the saved S and carries the running pair gradient G in place:
//     G_l = G_{l+1} + softmax'(S_l)    dq = scale * G_l k,  dk = scale * G_l^T q,  dv = Pd^T dO
#include "pair_attn.h"

namespace mmdti {

template <int NCH>
__global__ __launch_bounds__(256) void pair_attn_fwd_kernel(const bf16_t* __restrict__ qkv,
                                                            const float* __restrict__ bias_in,
                                                            float* __restrict__ s_out, bf16_t* __restrict__ o,
                                                            const unsigned char* __restrict__ key_pad, int N, int H,
                                                            int ld, float scale, uint32_t thresh, float dscale,
                                                            uint64_t seed, uint32_t site) {
  __shared__ __attribute__((aligned(16))) float sq[NCH * 64][HD];
  // (the 64 heads of a token share 128-byte q / k / v lines, 8 heads per line: keep a molecule's heads on one XCD)
  const int bh = xcd_chunk(blockIdx.x, gridDim.x), b = bh / H, h = bh - b * H;
  const int D = H * HD, D3 = 3 * D;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bf16_t* base = qkv + (long long)b * N * D3 + h * HD;
  for (int t = tid; t < N; t += 256) {
    float q[8];
    load8_bf16(base + (long long)t * D3, q);
#pragma unroll
    for (int d = 0; d < 8; ++d) sq[t][d] = q[d];
  }
  float k[NCH][8], v[NCH][8];
  bool masked[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int j = c * 64 + lane;
    masked[c] = true;
#pragma unroll
    for (int d = 0; d < 8; ++d) k[c][d] = v[c][d] = 0.f;
    if (j < N) {
      load8_bf16(base + (long long)j * D3 + D, k[c]);
      load8_bf16(base + (long long)j * D3 + 2 * D, v[c]);
      masked[c] = key_pad ? key_pad[b * N + j] != 0 : false;
    }
  }
  __syncthreads();
  const float NEG_INF = -INFINITY;
  // software pipeline over rows: the bias row of iteration i+4 is requested before row i is processed, so each wave keeps
  // two rows of pair traffic in flight (the stream is latency-bound otherwise: one row = 3 x 256 B per wave).
  float nb[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int j = c * 64 + lane;
    nb[c] = (wave < N && j < N && !masked[c]) ? bias_in[((long long)bh * N + wave) * ld + j] : 0.f;
  }
  for (int i = wave; i < N; i += 4) {
    const float4 q0 = *reinterpret_cast<const float4*>(&sq[i][0]);
    const float4 q1 = *reinterpret_cast<const float4*>(&sq[i][4]);
    const long long rowoff = ((long long)bh * N + i) * ld;
    float cb[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) cb[c] = nb[c];
    if (i + 4 < N) {
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int j = c * 64 + lane;
        nb[c] = (j < N && !masked[c]) ? bias_in[rowoff + 4LL * ld + j] : 0.f;
      }
    }
    float sv[NCH];
    float m = NEG_INF;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int j = c * 64 + lane;
      float s = NEG_INF;
      if (j < N) {
        float dot = q0.x * k[c][0] + q0.y * k[c][1] + q0.z * k[c][2] + q0.w * k[c][3] + q1.x * k[c][4] +
                    q1.y * k[c][5] + q1.z * k[c][6] + q1.w * k[c][7];
        s = masked[c] ? NEG_INF : scale * dot + cb[c];
        s_out[rowoff + j] = s;
      }
      sv[c] = s;
      m = fmaxf(m, s);
    }
    m = wave_max(m);
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      sv[c] = __expf(sv[c] - m);
      sum += sv[c];
    }
    const float inv = 1.0f / wave_sum(sum);
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      float p = sv[c] * inv;
      if (thresh) {
        const int j = c * 64 + lane;
        bool keep = dropout_keep(seed, site, ((uint64_t)bh * N + i) * ld + j, thresh);
        p = keep ? p * dscale : 0.f;
      }
#pragma unroll
      for (int d = 0; d < 8; ++d) acc[d] += p * v[c][d];
    }
    const float tot = wave_sum8_scatter(acc, lane);
    if (lane < 8) o[((long long)b * N + i) * D + h * HD + lane] = f2bf(tot);
  }
}

// RAG (ragged batches): key_tiles[b] = number of 16-key tiles of molecule b that hold a real key.  The tiles past it (past
// pa_kt_effective of it) are all padding -- -inf in every S of the chain, 0 in G -- so they are neither loaded nor computed nor
// (rag_store == 0) stored; pad QUERY rows are still computed (the reference's unmasked InfoNCE mean reads the encoder output at
// padded positions).  rag_store != 0: the skipped tiles are written as -inf (the last layer, whose S is returned to the caller).
template <int NT, bool TILED, bool FULL, bool RAG, typename ST, bool F16 = false>
__global__ __launch_bounds__(256, NT > 9 ? 2 : 4) void pair_attn_fwd_mfma_kernel(const bf16_t* __restrict__ qkv, const ST* __restrict__ bias_in,
                                                                 ST* __restrict__ s_out, bf16_t* __restrict__ o,
                                                                 const unsigned char* __restrict__ key_pad, int N, int H, int ld,
                                                                 float scale, uint32_t thresh, float dscale, uint64_t seed,
                                                                 uint32_t site, const int* __restrict__ key_tiles, int rag_store,
                                                                 const int* __restrict__ row_off) {
  static_assert(TILED || sizeof(ST) == 4, "compact pair tensors exist in the tiled layout only");
  static_assert(!RAG || sizeof(ST) == 2, "key-tile skipping is built for the compact layout only");
  constexpr int NP = NT * 16;
  constexpr int KSTR = NP + 8;   // row stride (elements) of the d-major V image (see the backward kernel's sKT)
  // raw bf16 images, exactly as loaded: sQ / sK [row][8], sVT [d][key]
  __shared__ __attribute__((aligned(16))) bf16_t sQ[NP * 8];
  __shared__ __attribute__((aligned(16))) bf16_t sK[NP * 8];
  __shared__ __attribute__((aligned(16))) bf16_t sVT[8 * KSTR];
  __shared__ __attribute__((aligned(16))) float sM[NP];
  // (the 64 heads of a token share 128-byte q / k / v lines, 8 heads per line: keep a molecule's heads on one XCD)
  const int bh = xcd_chunk(blockIdx.x, gridDim.x), b = bh / H, h = bh - b * H;
  const int D = H * HD, D3 = 3 * D;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
  const int nKB = (N + 15) >> 4;
  // (wave-uniform: one molecule per workgroup; rounded up to a count the sweeps are unrolled for)
  const int kt = RAG ? pa_kt_effective(min(__builtin_amdgcn_readfirstlane(key_tiles[b]), nKB), NT) : NT;
  // PACKED token rows (RAG only, row_off != null): molecule b owns rows [row_off[b], row_off[b+1]) of qkv / o / key_pad -- its real
  // tokens followed by at most ONE representative pad row (every pad row of a molecule is the same row at dropout 0: zeroed
  // input, constant bias row, same keys) -- instead of rows [b*N, (b+1)*N).  Pair planes stay indexed by position.
  const bool packed = RAG && row_off != nullptr;
  const int row0 = packed ? __builtin_amdgcn_readfirstlane(row_off[b]) : b * N;
  const int rows = packed ? __builtin_amdgcn_readfirstlane(row_off[b + 1]) - row0 : N;
  const bf16_t* base = qkv + (long long)row0 * D3 + h * HD;
  for (int t = tid; t < NP; t += blockDim.x) {
    uint4 q = make_uint4(0u, 0u, 0u, 0u), kk = q, vv = q;
    float msk = 1.f;
    if (t < rows) {
      q = *reinterpret_cast<const uint4*>(base + (long long)t * D3);
      kk = *reinterpret_cast<const uint4*>(base + (long long)t * D3 + D);
      vv = *reinterpret_cast<const uint4*>(base + (long long)t * D3 + 2 * D);
      msk = (key_pad && key_pad[row0 + t]) ? 1.f : 0.f;
    }
    *reinterpret_cast<uint4*>(sQ + t * 8) = q;
    *reinterpret_cast<uint4*>(sK + t * 8) = kk;
    const uint32_t vw[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
    for (int d = 0; d < 8; ++d) sVT[d * KSTR + t] = (bf16_t)((d & 1) ? (vw[d >> 1] >> 16) : (vw[d >> 1] & 0xffffu));
    sM[t] = msk;
  }
  for (int t = tid; t < 8 * 8; t += blockDim.x) sVT[(t >> 3) * KSTR + NP + (t & 7)] = 0;
  __syncthreads();
  const int g = lane >> 4, c16 = lane & 15;
  const float NEG_INF = -INFINITY;
  // Layouts (TILED): row-major [N][ld] planes, or the blocked-row planes of common.h whose 16x16 tiles are stored in accumulator
  // order -- a wave's access to a tile is then one contiguous KiB at a compile-time offset from the query block's base,
  // and, because every pad slot of a tiled tensor holds -inf (written by mmdti_gbf_bias_fwd, preserved by every S
  // store), interior tiles need no predicate and no pad masking at all.  Only the last query block (EDGE) and the last
  // key tile keep per-lane predicates.  FULL: nKB == NT, so "last tile" is a compile-time index.
  const int nlast = nKB - 1;
  const bool colok = 4 * g < N - 16 * nlast;
  const Rng24 rk = rng24_key(seed, site, (uint32_t)bh);     // dropout: the (molecule, head) plane's key (common.h)
  auto body = [&](int qb, auto edge_c, auto kt_c) {
    constexpr bool EDGE = decltype(edge_c)::value;
    constexpr int KT = decltype(kt_c)::value;   // key tiles this molecule's sweeps cover (NT unless RAG)
    const int qi = qb * 16 + c16;
    const uint32_t t8 = thresh ? rng24_row_t8(rk, (uint32_t)qi, thresh) : 0u;   // this query row's drop threshold
    const bool qvalid = EDGE ? qi < rows : true;
    const long long rowoff = ((long long)bh * N + (qvalid ? qi : 0)) * ld;   // (also the dropout counter base)
    const pa_s16x4 zero4 = {0, 0, 0, 0};
    // B of S^T = K.Q^T : Q[query c16][d = 4g..4g+3] (k = d: lane groups 2, 3 carry zeros)
    const pa_s16x4 qv = g < 2 ? *reinterpret_cast<const pa_s16x4*>(sQ + (qb * 16 + c16) * 8 + 4 * g) : zero4;
    // (tiled planes, common.h "blocked rows": the block of 16 queries starts at 16 qb N4, a lane's 4 keys of tile t sit at
    //  vr (16 t + 4 g) + 4 c16 -- 256 t + 4 lane in a complete block; lanes past the last query of an incomplete block point at query 0,
    //  key groups past N4 -- N <= 12 only -- at group 0: whatever a predicate-off lane reads lies inside the plane)
    const int vr = EDGE ? min(16, N - qb * 16) : 16;
    const long long tbase = (long long)bh * pair_plane(N) + (long long)qb * 16 * pair_n4(N) + ((4 * g < pair_n4(N) ? g : 0) * vr + (qvalid ? c16 : 0)) * 4;
    // (row-major: the lane's 4 keys of tile t start at rowoff + 16t + 4g; a lane whose predicate is off reads the row's
    //  first 16 bytes instead -- always inside the tensor, unlike "tile 0 + 4g" when ld < 16)
    const ST* bin = bias_in + (TILED ? tbase : rowoff);
    ST* sout = s_out + (TILED ? tbase : rowoff);
    const int TSTEP = TILED ? (EDGE ? vr * 16 : 256) : 16;
    const int goff = TILED ? 0 : 4 * g;
#define PA_PRED(T) (!TILED ? (qvalid && (T) * 16 + 4 * g < N) \
                           : (qvalid && (FULL ? ((T) < NT - 1 || colok) : ((T) < nlast || ((T) == nlast && colok)))))
    // interior tiles of a complete query block: no predicate (FULL only: as a wave-uniform run-time branch it doubles the unrolled
    // code and the NT = 13 backward fell out of the instruction cache, 1.2 -> 3.3 ms)
#define PA_FAST(T) (TILED && !EDGE && (FULL ? (T) < NT - 1 : false))
    // Phase 1: request every bias tile of this query block (the NT 16-byte loads are issued back to back and stay in
    // flight together).  Slots that are not loaded: -inf (pad keys), or 0 in a pad ROW (keeps that row's softmax finite;
    // nothing of it is stored).
    const float fillv = (TILED && qvalid) ? -INFINITY : 0.f;
    f32x4 S[KT];
#pragma unroll
    for (int t = 0; t < KT; ++t) {
      if (PA_FAST(t)) {
        S[t] = pa_load4_nt(bin + t * TSTEP + goff);
      } else {
        const bool pr = PA_PRED(t);
        const f32x4 ld4 = pa_load4(bin + (pr ? t * TSTEP + goff : 0));
        S[t] = pr ? ld4 : f32x4{fillv, fillv, fillv, fillv};
      }
    }
    float m = NEG_INF;
    if (RAG && rag_store) {   // the all-padding tiles past KT: -inf (the last layer: its S goes back to the caller)
      const f32x4 ninf = {NEG_INF, NEG_INF, NEG_INF, NEG_INF};
#pragma unroll
      for (int t = KT; t < NT; ++t) {
        if (PA_FAST(t)) {
          pa_store4_nt(sout + t * TSTEP + goff, ninf);
        } else if (PA_PRED(t)) {
          pa_store4_nt(sout + t * TSTEP + goff, ninf);
        }
      }
    }
#pragma unroll
    for (int t = 0; t < KT; ++t) {
      {
        const int kcol = t * 16 + 4 * g;
        f32x4 c = S[t];
        const pa_s16x4 ka = g < 2 ? *reinterpret_cast<const pa_s16x4*>(sK + (t * 16 + c16) * 8 + 4 * g) : zero4;
        f32x4 qk = {0.f, 0.f, 0.f, 0.f};
        qk = pa_mfma16<F16>(ka, qv, qk);   // exact products of the 16-bit operands, fp32 accumulation
#pragma unroll
        for (int r = 0; r < 4; ++r) c[r] += scale * qk[r];
        if (!TILED || key_pad) {   // (tiled tensors carry -inf in their pad keys already: only a real padding mask is left)
          const f32x4 km = *reinterpret_cast<const f32x4*>(&sM[kcol]);
#pragma unroll
          for (int r = 0; r < 4; ++r) c[r] = (km[r] != 0.f) ? NEG_INF : c[r];
        }
        typename std::conditional<sizeof(ST) == 2, pa_f16x4, f32x4>::type cs;
        c = pa_round4(cs, c);   // (compact: the fp16 value that is stored is also the one this layer's softmax sees)
        if (PA_FAST(t)) {
          pa_store4_nt(sout + t * TSTEP + goff, cs);
        } else if (PA_PRED(t)) {
          pa_store4_nt(sout + t * TSTEP + goff, cs);
        }
        S[t] = c;
        m = fmaxf(fmaxf(m, c[0]), c[1]);   // (two v_max3_f32)
        m = fmaxf(fmaxf(m, c[2]), c[3]);
      }
    }
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    // exp(S - m) as exp2(S * log2(e) - m * log2(e)): one FMA + v_exp_f32 per element (the backward recomputes it the same way)
    const float mneg = -m * PA_LOG2E;
    float lsum = 0.f;
#pragma unroll
    for (int t = 0; t < KT; ++t) {
      {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = __builtin_amdgcn_exp2f(fmaf(S[t][r], PA_LOG2E, mneg));
          S[t][r] = e;
          lsum += e;
        }
      }
    }
    lsum += __shfl_xor(lsum, 16, 64);
    lsum += __shfl_xor(lsum, 32, 64);
    const float inv = dscale / lsum;   // (dropout's 1 / (1 - p) folded in: dscale == 1 without dropout)
    f32x4 oacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < KT; ++t) {
      {
        f32x4 p = S[t] * inv;
        if (thresh) {
          const uint32_t kw = rng24_word(rk, ((uint32_t)qi * (uint32_t)ld + t * 16 + 4 * g) >> 2);
#pragma unroll
          for (int r = 0; r < 4; ++r) p[r] = rng24_kept(kw, r, t8) ? p[r] : 0.f;
        }
        // O^T += V^T . P^T : A = V[keys 16t + 4g..4g+3][d = c16 & 7] (rows d >= 8 of the result are never stored), B = P^T as it sits
        const pa_s16x4 va = *reinterpret_cast<const pa_s16x4*>(sVT + (c16 & 7) * KSTR + t * 16 + 4 * g);
        pa_s16x4 ph, pl;
        pa_split4_t<F16>(p, ph, pl);
        oacc = pa_mfma16<F16>(va, ph, oacc);
        oacc = pa_mfma16<F16>(va, pl, oacc);
      }
    }
    // O^T accumulator: column = query, rows d = 4g + r (valid for g < 2)
    if (qvalid && g < 2) {
      uint2 pk;
      if constexpr (F16) {   // (o feeds out_proj's forward GEMM: fp16 like every forward GEMM operand of this mode)
        pk.x = f2h_sat2(oacc[0], oacc[1]);
        pk.y = f2h_sat2(oacc[2], oacc[3]);
      } else {
        pk.x = (uint32_t)f2bf(oacc[0]) | ((uint32_t)f2bf(oacc[1]) << 16);
        pk.y = (uint32_t)f2bf(oacc[2]) | ((uint32_t)f2bf(oacc[3]) << 16);
      }
      *reinterpret_cast<uint2*>(o + ((long long)row0 + qi) * D + h * HD + 4 * g) = pk;
    }
#undef PA_PRED
#undef PA_FAST
  };
  const int nQB = packed ? (rows + 15) >> 4 : nKB;   // (packed: the pad rows past the representative one are not computed)
  auto run = [&](auto kt_c) {
    for (int qb = wave; qb < nQB; qb += nwaves) {
      if (TILED && qb * 16 + 16 <= rows) body(qb, std::false_type{}, kt_c);
      else body(qb, std::true_type{}, kt_c);
    }
  };
  if constexpr (RAG) pa_dispatch_kt<NT, NT>(kt, run);
  else run(std::integral_constant<int, NT>{});
}

// =====================================================================================================================
// MFMA-tiled backward, same tiling as the forward: a wave owns 16 queries and sweeps the key tiles twice.
//   sweep 1: load S^T tiles, row max / sum  ->  P^T ; dP^T = V.dO^T (MFMA) ; dropout mask (kept in the sign of P) ;
//            delta = rowsum(dP' * P)
//   sweep 2: G^T = G_in^T + P^T*(dP'^T - delta)  (16-byte load + store per lane per tile, in place) ;
//            dQ^T += K^T.G^T (MFMA, G^T is already the B operand) ;
//            dK += G.Q and dV += Pd.dO contract over QUERIES (the lane axis of the accumulator layout), so each tile is
//            transposed through a per-wave 16x16 LDS patch: written row-major, read back with ds_read_b64_tr_b16.
// Precision of sweep 2: Q, K, V, dO are bf16 in memory, so they enter the MFMAs exactly; the fp32 factors G and
// dropout(P) are split into bf16 high + bf16 low parts (x = hi + lo + O(2^-17 |x|)) and every product runs twice on
// v_mfma_f32_16x16x16_bf16 -- fp32-class products (16 mantissa bits kept of the fp32 factor, fp32 accumulation) at a
// quarter of the matrix-pipe time and a third of the LDS instructions of the fp32 16x16x4 form this replaces.
// Each (query block, key tile) contribution to dK/dV is added to an LDS image owned by the wave.

}  // namespace mmdti
MMDTI_DEFINE_SALT_PULL(pair_attn)
using namespace mmdti;

extern "C" int mmdti_pair_attn_fwd(mmdti_stream_t stream, const void* qkv_bf16, const void* bias_in, void* s_out,
                                   void* o_bf16, const unsigned char* key_pad, int B, int N, int H, int ld,
                                   float scale, float drop_p, unsigned long long seed, unsigned int site, int layout,
                                   const int* key_tiles, int rag_store, const int* row_off, int qkv_f16) {
  const int tiled = layout & 1, compact = (layout >> 1) & 1;   // bit 0: tiled planes; bit 1: compact planes (S fp16; tiled only)
  if (int e = check_common("pair_attn_fwd", B, N, H, ld)) return e;
  MMDTI_REQUIRE((layout & ~3) == 0 && (!compact || tiled), "pair_attn_fwd: layout must be 0 (row-major fp32), 1 (tiled fp32) or 3 (tiled, fp16 logits)");
  MMDTI_REQUIRE(!key_tiles || compact, "pair_attn_fwd: key_tiles (ragged batches) needs the compact tiled pair layout (layout 3)");
  MMDTI_REQUIRE(!row_off || key_tiles, "pair_attn_fwd: packed token rows (row_off) need key_tiles");
  MMDTI_REQUIRE(!qkv_f16 || compact, "pair_attn_fwd: fp16 q | k | v (the fp16 forward-operand mode) is built for the compact tiled pair layout (layout 3)");
  MMDTI_REQUIRE(!tiled || (ld % 4 == 0 && N <= 16 * PA_MAX_NT), "pair_attn_fwd: the tiled pair layout needs ld %% 4 == 0 and N <= 272");
  MMDTI_REQUIRE(qkv_bf16 && bias_in && s_out && o_bf16, "pair_attn_fwd: null pointer");
  MMDTI_REQUIRE(aligned16(qkv_bf16), "pair_attn_fwd: qkv must be 16-byte aligned");
  MMDTI_REQUIRE(!tiled || (aligned16(bias_in) && aligned16(s_out)), "pair_attn_fwd: tiled pair tensors must be 16-byte aligned");
  MMDTI_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "pair_attn_fwd: dropout p out of range");
  const uint32_t th = dropout_thresh(drop_p);     // the per-element kernel (N > 272 or unaligned rows): Philox words against p * 2^32
  const uint32_t th8 = dropout_thresh8(drop_p);   // the MFMA kernels: the shared byte generator of common.h
  const float sc = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  dim3 grid(B * H), block(256);
  hipStream_t s = (hipStream_t)stream;
  // (same eligibility rule as the backward: the two MFMA kernels share one dropout-mask generator)
  if (ld % 4 == 0 && aligned16(bias_in) && aligned16(s_out) && N <= 16 * PA_MAX_NT) {
    const int nqb = (N + 15) / 16;
    dim3 blk(nqb % 3 == 0 ? 192 : (nqb < 4 ? 64 * nqb : 256));
#define PA_M(NT, TL, FL, RG, ST)                                                                                                     \
  hipLaunchKernelGGL((pair_attn_fwd_mfma_kernel<NT, TL, FL, RG, ST>), grid, blk, 0, s, (const bf16_t*)qkv_bf16, (const ST*)bias_in, \
                     (ST*)s_out, (bf16_t*)o_bf16, key_pad, N, H, ld, scale, th8, sc, (uint64_t)seed, (uint32_t)site, key_tiles, rag_store, row_off)
#define PA_MT(NT)                                                                           \
  do {                                                                                      \
    if (!tiled) PA_M(NT, false, false, false, float);                                       \
    else if (!compact) { if (nqb == NT) PA_M(NT, true, true, false, float); else PA_M(NT, true, false, false, float); } \
  } while (0)
    // The hot path (compact planes) has an instantiation for EVERY tile count: all its tiles exist, the interior ones run
    // without predicates (FULL).  In a kernel instantiated for more tiles than N has, the surplus tiles are predicated off but
    // still walked and no tile takes the fast path -- 25-45 % more time per real tile (N = 96 / 113 / 128 on the 9-tile kernel).
#define PA_MH(NT, RG)                                                                                                                  \
  hipLaunchKernelGGL((pair_attn_fwd_mfma_kernel<NT, true, true, RG, _Float16, true>), grid, blk, 0, s, (const bf16_t*)qkv_bf16,       \
                     (const _Float16*)bias_in, (_Float16*)s_out, (bf16_t*)o_bf16, key_pad, N, H, ld, scale, th8, sc, (uint64_t)seed,  \
                     (uint32_t)site, key_tiles, rag_store, row_off)
#define PA_MC(NT) case NT: if (qkv_f16) { if (key_tiles) PA_MH(NT, true); else PA_MH(NT, false); } \
                           else if (key_tiles) PA_M(NT, true, true, true, _Float16); else PA_M(NT, true, true, false, _Float16); break
    if (compact) {
      switch (nqb) {
        PA_MC(1); PA_MC(2); PA_MC(3); PA_MC(4); PA_MC(5); PA_MC(6); PA_MC(7); PA_MC(8); PA_MC(9); PA_MC(10); PA_MC(11); PA_MC(12);
        PA_MC(13); PA_MC(14); PA_MC(15); PA_MC(16); PA_MC(17);
      }
    } else if (nqb <= 5) PA_MT(5); else if (nqb <= 9) PA_MT(9); else if (nqb <= 13) PA_MT(13); else PA_MT(17);
#undef PA_MC
#undef PA_MH
#undef PA_MT
#undef PA_M
    MMDTI_LAUNCH_CHECK();
    return MMDTI_OK;
  }
#define PA_F(NCH)                                                                                                   \
  hipLaunchKernelGGL((pair_attn_fwd_kernel<NCH>), grid, block, 0, s, (const bf16_t*)qkv_bf16, (const float*)bias_in, (float*)s_out, \
                     (bf16_t*)o_bf16, key_pad, N, H, ld, scale, th, sc, (uint64_t)seed, (uint32_t)site)
  switch ((N + 63) / 64) {
    case 1: PA_F(1); break;
    case 2: PA_F(2); break;
    case 3: PA_F(3); break;
    case 4: PA_F(4); break;
    default: PA_F(5); break;
  }
#undef PA_F
  MMDTI_LAUNCH_CHECK();
  return MMDTI_OK;
}
