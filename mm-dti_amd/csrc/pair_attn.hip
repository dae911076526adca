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
#include <type_traits>

#include "common.h"

namespace mmdti {

constexpr int HD = 8;

__device__ __forceinline__ void load8_bf16(const bf16_t* p, float (&o)[8]) {
  uint4 u = *reinterpret_cast<const uint4*>(p);
  uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    o[2 * i] = __uint_as_float(w[i] << 16);
    o[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
  }
}
__device__ __forceinline__ void store8_bf16(bf16_t* p, const float (&v)[8]) {
  uint4 u;
  u.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
  u.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
  u.z = (uint32_t)f2bf(v[4]) | ((uint32_t)f2bf(v[5]) << 16);
  u.w = (uint32_t)f2bf(v[6]) | ((uint32_t)f2bf(v[7]) << 16);
  *reinterpret_cast<uint4*>(p) = u;
}

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

template <int NCH>
__global__ __launch_bounds__(256) void pair_attn_bwd_kernel(const bf16_t* __restrict__ qkv, const float* __restrict__ s,
                                                            const bf16_t* __restrict__ dO, float* __restrict__ g,
                                                            bf16_t* __restrict__ dqkv, int N, int H, int ld,
                                                            float scale, int g_in_zero, uint32_t thresh, float dscale,
                                                            uint64_t seed, uint32_t site) {
  __shared__ __attribute__((aligned(16))) float sq[NCH * 64][HD];
  __shared__ __attribute__((aligned(16))) float sdo[NCH * 64][HD];
  __shared__ __attribute__((aligned(16))) float red[NCH * 64][2 * HD];
  // (the 64 heads of a token share 128-byte q / k / v lines, 8 heads per line: keep a molecule's heads on one XCD)
  const int bh = xcd_chunk(blockIdx.x, gridDim.x), b = bh / H, h = bh - b * H;
  const int D = H * HD, D3 = 3 * D;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bf16_t* base = qkv + (long long)b * N * D3 + h * HD;
  for (int t = tid; t < NCH * 64; t += 256) {
    float q[8] = {0, 0, 0, 0, 0, 0, 0, 0}, d_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (t < N) {
      load8_bf16(base + (long long)t * D3, q);
      load8_bf16(dO + ((long long)b * N + t) * D + h * HD, d_);
    }
#pragma unroll
    for (int d = 0; d < 8; ++d) {
      sq[t][d] = q[d];
      sdo[t][d] = d_[d];
      red[t][d] = 0.f;
      red[t][8 + d] = 0.f;
    }
  }
  float k[NCH][8], v[NCH][8], dk[NCH][8], dv[NCH][8];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int j = c * 64 + lane;
#pragma unroll
    for (int d = 0; d < 8; ++d) k[c][d] = v[c][d] = dk[c][d] = dv[c][d] = 0.f;
    if (j < N) {
      load8_bf16(base + (long long)j * D3 + D, k[c]);
      load8_bf16(base + (long long)j * D3 + 2 * D, v[c]);
    }
  }
  __syncthreads();
  const float NEG_INF = -INFINITY;
  // software pipeline over rows (see the forward kernel): S and G of row i+4 are in flight while row i is processed
  float ns[NCH], ng[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int j = c * 64 + lane;
    const bool ok = wave < N && j < N;
    ns[c] = ok ? s[((long long)bh * N + wave) * ld + j] : NEG_INF;
    ng[c] = (ok && !g_in_zero) ? g[((long long)bh * N + wave) * ld + j] : 0.f;
  }
  for (int i = wave; i < N; i += 4) {
    float q[8], dd[8];
    {
      const float4 a0 = *reinterpret_cast<const float4*>(&sq[i][0]), a1 = *reinterpret_cast<const float4*>(&sq[i][4]);
      const float4 b0 = *reinterpret_cast<const float4*>(&sdo[i][0]), b1 = *reinterpret_cast<const float4*>(&sdo[i][4]);
      q[0] = a0.x; q[1] = a0.y; q[2] = a0.z; q[3] = a0.w; q[4] = a1.x; q[5] = a1.y; q[6] = a1.z; q[7] = a1.w;
      dd[0] = b0.x; dd[1] = b0.y; dd[2] = b0.z; dd[3] = b0.w; dd[4] = b1.x; dd[5] = b1.y; dd[6] = b1.z; dd[7] = b1.w;
    }
    const long long rowoff = ((long long)bh * N + i) * ld;
    float p[NCH], dpp[NCH], pd[NCH], gin[NCH];
    float m = NEG_INF;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      p[c] = ns[c];
      gin[c] = ng[c];
      m = fmaxf(m, p[c]);
    }
    if (i + 4 < N) {
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int j = c * 64 + lane;
        ns[c] = (j < N) ? s[rowoff + 4LL * ld + j] : NEG_INF;
        ng[c] = (j < N && !g_in_zero) ? g[rowoff + 4LL * ld + j] : 0.f;
      }
    }
    m = wave_max(m);
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      p[c] = __expf(p[c] - m);
      sum += p[c];
    }
    const float inv = 1.0f / wave_sum(sum);
    float dl = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      p[c] *= inv;
      float dp = 0.f;
#pragma unroll
      for (int d = 0; d < 8; ++d) dp += dd[d] * v[c][d];
      float keepscale = 1.f;
      if (thresh) {
        const int j = c * 64 + lane;
        keepscale = dropout_keep(seed, site, ((uint64_t)bh * N + i) * ld + j, thresh) ? dscale : 0.f;
      }
      dpp[c] = dp * keepscale;
      pd[c] = p[c] * keepscale;
      dl += dpp[c] * p[c];
    }
    dl = wave_sum(dl);
    float dq[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int j = c * 64 + lane;
      float gg = 0.f;
      if (j < N) {
        gg = p[c] * (dpp[c] - dl) + gin[c];
        g[rowoff + j] = gg;
      }
#pragma unroll
      for (int d = 0; d < 8; ++d) {
        dq[d] += gg * k[c][d];
        dk[c][d] += gg * q[d];
        dv[c][d] += pd[c] * dd[d];
      }
    }
    const float tot = wave_sum8_scatter(dq, lane) * scale;
    if (lane < 8) dqkv[((long long)b * N + i) * D3 + h * HD + lane] = f2bf(tot);
  }
  // combine the 4 waves' dK / dV partials through LDS atomics
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int j = c * 64 + lane;
    if (j < N) {
#pragma unroll
      for (int d = 0; d < 8; ++d) {
        atomicAdd(&red[j][d], dk[c][d] * scale);
        atomicAdd(&red[j][8 + d], dv[c][d]);
      }
    }
  }
  __syncthreads();
  for (int t = tid; t < N; t += 256) {
    float a[8], c2[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) {
      a[d] = red[t][d];
      c2[d] = red[t][8 + d];
    }
    bf16_t* dst = dqkv + ((long long)b * N + t) * D3 + h * HD;
    store8_bf16(dst + D, a);
    store8_bf16(dst + 2 * D, c2);
  }
}

// =====================================================================================================================
// MFMA-tiled forward.  One wave owns a block of 16 queries and walks the key tiles of 16: the score tile is computed
// TRANSPOSED, S^T[key][query] = K.Q^T (one v_mfma_f32_16x16x16_bf16 on the raw bf16 q / k rows: head_dim 8 zero-padded
// to k = 16, exact products, fp32 accumulation), scaled and added to the bias tile in the accumulator layout, so bias
// add, S write-out and softmax all happen on that layout
// (lane = query column, 4 consecutive keys per lane-group in registers): pair traffic is one 16-byte load + one 16-byte
// store per lane per tile, the row softmax needs only two cross-lane steps per query block, and P^T is already the B
// operand of the P.V product (O^T = V^T.P^T) -- no data movement between the two matrix products.
// Dropout element index = (bh*N + query)*ld + key (ld % 4 == 0: one RNG call per 4 keys).
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Matrix products of both MFMA kernels: Q, K, V, dO are bf16 in memory, so they enter v_mfma_f32_16x16x16_bf16 exactly as
// loaded; an fp32 factor (the probabilities, G) is split into bf16 high + bf16 low parts (x = hi + lo + O(2^-17 |x|)) and
// its product runs twice -- fp32-class products with fp32 accumulation at a quarter of the matrix-pipe time and a third
// of the LDS instructions of the fp32 16x16x4 form.
typedef short pa_s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 pa_bf16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) pa_s16x4 pa_lds_s16x4;

__device__ __forceinline__ pa_s16x4 pa_pack4(const f32x4& v) {
  pa_bf16x4 h;
  h[0] = (__bf16)v[0]; h[1] = (__bf16)v[1]; h[2] = (__bf16)v[2]; h[3] = (__bf16)v[3];
  return __builtin_bit_cast(pa_s16x4, h);
}
// x = hi + lo with hi = bf16(x), lo = bf16(x - hi)
__device__ __forceinline__ void pa_split4(const f32x4& v, pa_s16x4& hi, pa_s16x4& lo) {
  pa_bf16x4 h;
  h[0] = (__bf16)v[0]; h[1] = (__bf16)v[1]; h[2] = (__bf16)v[2]; h[3] = (__bf16)v[3];
  f32x4 r;
  r[0] = v[0] - (float)h[0]; r[1] = v[1] - (float)h[1]; r[2] = v[2] - (float)h[2]; r[3] = v[3] - (float)h[3];
  hi = __builtin_bit_cast(pa_s16x4, h);
  lo = pa_pack4(r);
}
#define PA_MFMA16(A, B, C) __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(A, B, C, 0, 0, 0)
#define PA_LOG2E 1.44269504088896340736f

// Element types of the pair tensors.  Row-major planes and the round-1 tiled planes hold fp32.  The COMPACT tiled planes
// hold the logits chain S as fp16 (same [nKB][nKB][256] element order: a lane's 4 keys are 8 contiguous bytes, a tile
// 512 B): the pair kernels are bound by this traffic -- the forward's halves, the backward's drops from 12 to 10 B per
// pair and head.  The gradient chain G stays fp32 by default; bf16 is an opt-in (layout bit 2) that costs gradient
// fidelity where sums over pairs cancel ("tried").
//   S (fp16, round to nearest even): what the reference's own AMP path carries between layers (its attn_weights are fp16
//     under autocast); the softmax of the layer that produced a value runs on the ROUNDED value, so the forward and the
//     backward's recomputation see the same logits.  Values above the fp16 range saturate at 65504 instead of becoming
//     +inf (a -inf row mask stays -inf).
//   G (bf16, opt-in): fp16 would flush the small gradients the reference protects with its GradScaler; bf16 keeps fp32's
//     range.  The dQ / dK products of a layer use the unrounded fp32 G of that layer; only what is handed to the previous
//     layer is rounded.
typedef _Float16 pa_f16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 pa_load4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 pa_load4_nt(const float* p) { return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p)); }
__device__ __forceinline__ void pa_store4_nt(float* p, f32x4 v) { __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p)); }
__device__ __forceinline__ f32x4 pa_round4(f32x4& keep, f32x4 v) { keep = v; return v; }

__device__ __forceinline__ f32x4 pa_widen_f16(const pa_f16x4& h) {
  f32x4 v;
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = (float)h[r];
  return v;
}
__device__ __forceinline__ pa_f16x4 pa_narrow_f16(const f32x4& v) {
  pa_f16x4 h;
#pragma unroll
  for (int r = 0; r < 4; ++r) h[r] = (_Float16)v[r];
  return h;
}
__device__ __forceinline__ f32x4 pa_load4(const _Float16* p) { return pa_widen_f16(*reinterpret_cast<const pa_f16x4*>(p)); }
__device__ __forceinline__ f32x4 pa_load4_nt(const _Float16* p) { return pa_widen_f16(__builtin_nontemporal_load(reinterpret_cast<const pa_f16x4*>(p))); }
__device__ __forceinline__ void pa_store4_nt(_Float16* p, const f32x4& v) {   // (v already went through pa_round4: the conversion is exact)
  __builtin_nontemporal_store(pa_narrow_f16(v), reinterpret_cast<pa_f16x4*>(p));
}
// round a quad of logits to its storage type; `keep` receives the stored form so that the store does not convert again.
// (v_med3_f32 against (-inf, 65504) is min(v, 65504) without fminf's NaN-canonicalising v_max_f32 in front)
__device__ __forceinline__ f32x4 pa_round4(pa_f16x4& keep, const f32x4& v) {
  f32x4 c;
#pragma unroll
  for (int r = 0; r < 4; ++r) c[r] = __builtin_amdgcn_fmed3f(v[r], -INFINITY, 65504.f);
  keep = pa_narrow_f16(c);
  return pa_widen_f16(keep);
}
__device__ __forceinline__ void pa_store4_nt(_Float16* p, const pa_f16x4& h) { __builtin_nontemporal_store(h, reinterpret_cast<pa_f16x4*>(p)); }

__device__ __forceinline__ f32x4 pa_widen_bf16(const pa_s16x4& h) {
  f32x4 v;
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = __builtin_bit_cast(float, (uint32_t)(uint16_t)h[r] << 16);
  return v;
}
__device__ __forceinline__ f32x4 pa_load4(const __bf16* p) { return pa_widen_bf16(*reinterpret_cast<const pa_s16x4*>(p)); }
__device__ __forceinline__ f32x4 pa_load4_nt(const __bf16* p) { return pa_widen_bf16(__builtin_nontemporal_load(reinterpret_cast<const pa_s16x4*>(p))); }
__device__ __forceinline__ void pa_store4_nt(__bf16* p, const f32x4& v) { __builtin_nontemporal_store(pa_pack4(v), reinterpret_cast<pa_s16x4*>(p)); }

// Ragged batches: the number of key tiles a molecule's sweeps cover is a COMPILE-TIME constant of the code that runs them.  A
// workgroup reads its molecule's count and branches ONCE into the body unrolled for it; per-tile `if (t >= kt) skip` branches
// inside one body unrolled for all NT tiles were measured at + 20-40 % per processed tile (they keep the loads of later tiles
// from being issued ahead).  Supported counts: every k up to 9 tiles, every 2nd up to 13, every 4th beyond, and NT itself -- a
// count in between runs as the next supported one (the extra tiles are ordinary all-padding tiles: -inf logits, zero gradient,
// read and written like any other), identically in every layer and in both directions.
__host__ __device__ constexpr int pa_kt_step(int nt) { return nt <= 9 ? 1 : (nt <= 13 ? 2 : 4); }
// smallest supported count >= kt
__host__ __device__ constexpr int pa_kt_effective(int kt, int nt) {
  const int s = pa_kt_step(nt), k = ((kt < 1 ? 1 : kt) + s - 1) / s * s;
  return k >= nt ? nt : k;
}
// f(std::integral_constant<int, ke>) for a count ke that pa_kt_effective produced (K walks down the supported counts)
template <int NT, int K, typename F>
__device__ __forceinline__ void pa_dispatch_kt(int ke, F&& f) {
  constexpr int step = pa_kt_step(NT);
  constexpr int below = (K == NT) ? ((NT - 1) / step) * step : K - step;   // the next supported count below K (0: none)
  if constexpr (below < 1) {
    f(std::integral_constant<int, K>{});
  } else {
    if (ke >= K) f(std::integral_constant<int, K>{});
    else pa_dispatch_kt<NT, below>(ke, f);
  }
}

// RAG (ragged batches): key_tiles[b] = number of 16-key tiles of molecule b that hold a real key.  The tiles past it (past
// pa_kt_effective of it) are all padding -- -inf in every S of the chain, 0 in G -- so they are neither loaded nor computed nor
// (rag_store == 0) stored; pad QUERY rows are still computed (the reference's unmasked InfoNCE mean reads the encoder output at
// padded positions).  rag_store != 0: the skipped tiles are written as -inf (the last layer, whose S is returned to the caller).
template <int NT, bool TILED, bool FULL, bool RAG, typename ST>
__global__ __launch_bounds__(256, NT > 9 ? 2 : 4) void pair_attn_fwd_mfma_kernel(const bf16_t* __restrict__ qkv, const ST* __restrict__ bias_in,
                                                                 ST* __restrict__ s_out, bf16_t* __restrict__ o,
                                                                 const unsigned char* __restrict__ key_pad, int N, int H, int ld,
                                                                 float scale, uint32_t thresh, float dscale, uint64_t seed,
                                                                 uint32_t site, const int* __restrict__ key_tiles, int rag_store) {
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
  const bf16_t* base = qkv + (long long)b * N * D3 + h * HD;
  for (int t = tid; t < NP; t += blockDim.x) {
    uint4 q = make_uint4(0u, 0u, 0u, 0u), kk = q, vv = q;
    float msk = 1.f;
    if (t < N) {
      q = *reinterpret_cast<const uint4*>(base + (long long)t * D3);
      kk = *reinterpret_cast<const uint4*>(base + (long long)t * D3 + D);
      vv = *reinterpret_cast<const uint4*>(base + (long long)t * D3 + 2 * D);
      msk = (key_pad && key_pad[b * N + t]) ? 1.f : 0.f;
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
  // Layouts (TILED): row-major [N][ld] planes, or [nKB][nKB][256] planes whose 16x16 tiles are stored in accumulator
  // order -- a wave's access to a tile is then one contiguous KiB at a compile-time offset from the query block's base,
  // and, because every pad slot of a tiled tensor holds -inf (written by mmdti_gbf_bias_fwd, preserved by every S
  // store), interior tiles need no predicate and no pad masking at all.  Only the last query block (EDGE) and the last
  // key tile keep per-lane predicates.  FULL: nKB == NT, so "last tile" is a compile-time index.
  const int nlast = nKB - 1;
  const bool colok = 4 * g < N - 16 * nlast;
  auto body = [&](int qb, auto edge_c, auto kt_c) {
    constexpr bool EDGE = decltype(edge_c)::value;
    constexpr int KT = decltype(kt_c)::value;   // key tiles this molecule's sweeps cover (NT unless RAG)
    const int qi = qb * 16 + c16;
    const bool qvalid = EDGE ? qi < N : true;
    const long long rowoff = ((long long)bh * N + (qvalid ? qi : 0)) * ld;   // (also the dropout counter base)
    const pa_s16x4 zero4 = {0, 0, 0, 0};
    // B of S^T = K.Q^T : Q[query c16][d = 4g..4g+3] (k = d: lane groups 2, 3 carry zeros)
    const pa_s16x4 qv = g < 2 ? *reinterpret_cast<const pa_s16x4*>(sQ + (qb * 16 + c16) * 8 + 4 * g) : zero4;
    const long long tbase = ((long long)bh * nKB + qb) * nKB * 256 + lane * 4;
    // (row-major: the lane's 4 keys of tile t start at rowoff + 16t + 4g; a lane whose predicate is off reads the row's
    //  first 16 bytes instead -- always inside the tensor, unlike "tile 0 + 4g" when ld < 16)
    const ST* bin = bias_in + (TILED ? tbase : rowoff);
    ST* sout = s_out + (TILED ? tbase : rowoff);
    constexpr int TSTEP = TILED ? 256 : 16;
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
        qk = PA_MFMA16(ka, qv, qk);   // exact bf16 products, fp32 accumulation
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
          const Keep4 kw = keep4_words(seed, site, (uint64_t)(rowoff + t * 16 + 4 * g) >> 2);
#pragma unroll
          for (int r = 0; r < 4; ++r) p[r] = keep4_kept(kw, r, thresh >> 16) ? p[r] : 0.f;
        }
        // O^T += V^T . P^T : A = V[keys 16t + 4g..4g+3][d = c16 & 7] (rows d >= 8 of the result are never stored), B = P^T as it sits
        const pa_s16x4 va = *reinterpret_cast<const pa_s16x4*>(sVT + (c16 & 7) * KSTR + t * 16 + 4 * g);
        pa_s16x4 ph, pl;
        pa_split4(p, ph, pl);
        oacc = PA_MFMA16(va, ph, oacc);
        oacc = PA_MFMA16(va, pl, oacc);
      }
    }
    // O^T accumulator: column = query, rows d = 4g + r (valid for g < 2)
    if (qvalid && g < 2) {
      uint2 pk;
      pk.x = (uint32_t)f2bf(oacc[0]) | ((uint32_t)f2bf(oacc[1]) << 16);
      pk.y = (uint32_t)f2bf(oacc[2]) | ((uint32_t)f2bf(oacc[3]) << 16);
      *reinterpret_cast<uint2*>(o + ((long long)b * N + qi) * D + h * HD + 4 * g) = pk;
    }
#undef PA_PRED
#undef PA_FAST
  };
  auto run = [&](auto kt_c) {
    for (int qb = wave; qb < nKB; qb += nwaves) {
      if (TILED && qb * 16 + 16 <= N) body(qb, std::false_type{}, kt_c);
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

// RAG: see the forward kernel.  Skipped key tiles contribute nothing (P = 0, G = 0) and their G is NOT written: the caller
// hands in a zero-initialised G when the batch is ragged.
template <int NT, bool TILED, bool FULL, int NW, bool RAG, typename ST, typename GT>
__global__ __launch_bounds__(64 * NW, NT > 9 ? 1 : 3) void pair_attn_bwd_mfma_kernel(const bf16_t* __restrict__ qkv, const ST* __restrict__ s_in,
                                                                 const bf16_t* __restrict__ dO, const GT* __restrict__ gin, GT* __restrict__ gout,
                                                                 bf16_t* __restrict__ dqkv, int N, int H, int ld, float scale,
                                                                 int g_in_zero, uint32_t thresh, float dscale, uint64_t seed,
                                                                 uint32_t site, const int* __restrict__ key_tiles) {
  static_assert(TILED || (sizeof(ST) == 4 && sizeof(GT) == 4), "compact pair tensors exist in the tiled layout only");
  static_assert(!RAG || sizeof(ST) == 2, "key-tile skipping is built for the compact layout only");
  constexpr int NP = NT * 16;
  constexpr int KSTR = NP + 8;   // row stride (elements) of the d-major K image: 8-byte reads of 8 rows x 2 key groups hit 16 distinct bank pairs
  // raw bf16 images, exactly as loaded.  sQ / sD / sV: [row][8]; the +16 elements are the tail that tr-reads of the last
  // rows run into (they only feed output columns d >= 8, which are never used).  sKT: [d][key].
  __shared__ __attribute__((aligned(16))) bf16_t sQ[NP * 8 + 16];
  __shared__ __attribute__((aligned(16))) bf16_t sD[NP * 8 + 16];   // dO
  __shared__ __attribute__((aligned(16))) bf16_t sV[NP * 8];
  __shared__ __attribute__((aligned(16))) bf16_t sKT[8 * KSTR];
  // per-wave dK / dV accumulators in MFMA accumulator order: [wave][tile][K|V][g][d][r] -- each lane owns one float4 per
  // (tile, K|V), read as the MFMA C input and written back, so the waves never contend (LDS float atomics cost ~57
  // cycles per wave-instruction here and were half of the kernel's time).
  __shared__ __attribute__((aligned(16))) float redw[NW * NT * 2 * 128];
  // per-wave transpose patches [P | G], each [16 queries][16 keys] bf16 (512 B), used twice per tile (high parts, then low
  // parts: with two patches instead of four the workgroup stays under 40 KB -> four per CU); the 8-byte slot s of row q
  // sits at slot s ^ (2 * (q >> 3)), which makes both the row writes and the transposing reads conflict-free
  __shared__ __attribute__((aligned(16))) bf16_t patch[NW][2][256];
  // (the 64 heads of a token share 128-byte q / k / v lines, 8 heads per line: keep a molecule's heads on one XCD)
  const int bh = xcd_chunk(blockIdx.x, gridDim.x), b = bh / H, h = bh - b * H;
  const int D = H * HD, D3 = 3 * D;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
  const int nKB = (N + 15) >> 4;
  const int kt = RAG ? pa_kt_effective(min(__builtin_amdgcn_readfirstlane(key_tiles[b]), nKB), NT) : NT;   // (see the forward kernel)
  const bf16_t* base = qkv + (long long)b * N * D3 + h * HD;
  for (int t = tid; t < NP + 2; t += blockDim.x) {
    uint4 q = make_uint4(0u, 0u, 0u, 0u), kk = q, vv = q, dd = q;
    if (t < N) {
      q = *reinterpret_cast<const uint4*>(base + (long long)t * D3);
      kk = *reinterpret_cast<const uint4*>(base + (long long)t * D3 + D);
      vv = *reinterpret_cast<const uint4*>(base + (long long)t * D3 + 2 * D);
      dd = *reinterpret_cast<const uint4*>(dO + ((long long)b * N + t) * D + h * HD);
    }
    *reinterpret_cast<uint4*>(sQ + t * 8) = q;      // (t = NP, NP + 1: the zeroed tails)
    *reinterpret_cast<uint4*>(sD + t * 8) = dd;
    if (t < NP) {
      *reinterpret_cast<uint4*>(sV + t * 8) = vv;
      const uint32_t kw[4] = {kk.x, kk.y, kk.z, kk.w};
#pragma unroll
      for (int d = 0; d < 8; ++d) sKT[d * KSTR + t] = (bf16_t)((d & 1) ? (kw[d >> 1] >> 16) : (kw[d >> 1] & 0xffffu));
    }
  }
  for (int t = tid; t < 8 * 8; t += blockDim.x) sKT[(t >> 3) * KSTR + NP + (t & 7)] = 0;
  for (int t = tid; t < NW * NT * 2 * 128; t += blockDim.x) redw[t] = 0.f;
  __syncthreads();
  const int g = lane >> 4, c16 = lane & 15;
  const bool dlane = c16 < 8;
  const float NEG_INF = -INFINITY;
  bf16_t* pw = &patch[wave][0][0];
  // patch addressing (elements): this lane WRITES row c16, logical slot g ; tr-READS address row 4g + (c16 >> 2), logical slot c16 & 3
  const int pwr = c16 * 16 + ((g ^ ((c16 >> 3) << 1)) << 2);
  const int prd = (4 * g + (c16 >> 2)) * 16 + (((c16 & 3) ^ ((g >> 1) << 1)) << 2);
  const pa_s16x4 zero4 = {0, 0, 0, 0};
  // (layouts, TILED / FULL / EDGE: see the forward kernel.  In a tiled S every pad slot is -inf and in a tiled G every pad
  //  slot is 0 -- both are preserved by the stores below --, so interior tiles run without predicates or pad masking.)
  const int nlast = nKB - 1;
  const bool colok = 4 * g < N - 16 * nlast;
  auto body = [&](int qb, auto edge_c, auto kt_c) {
    constexpr bool EDGE = decltype(edge_c)::value;
    constexpr int KT = decltype(kt_c)::value;   // key tiles this molecule's sweeps cover (NT unless RAG)
    const int qi = qb * 16 + c16;
    const bool qvalid = EDGE ? qi < N : true;
    const long long rowoff = ((long long)bh * N + (qvalid ? qi : 0)) * ld;   // (also the dropout counter base)
    const long long tbase = ((long long)bh * nKB + qb) * nKB * 256 + lane * 4;
    const ST* sin_p = s_in + (TILED ? tbase : rowoff);     // (predicate-off lanes read the row's first 16 bytes: in bounds)
    const GT* gin_p = gin + (TILED ? tbase : rowoff);
    GT* gout_p = gout + (TILED ? tbase : rowoff);
    constexpr int TSTEP = TILED ? 256 : 16;
    const int goff = TILED ? 0 : 4 * g;
#define PA_PRED(T) (!TILED ? (qvalid && (T) * 16 + 4 * g < N) \
                           : (qvalid && (FULL ? ((T) < NT - 1 || colok) : ((T) < nlast || ((T) == nlast && colok)))))
#define PA_FAST(T) (TILED && !EDGE && (FULL ? (T) < NT - 1 : false))
    // operands of this query block that do not depend on the key tile:
    //   dob : B of dP^T = V.dO^T          -> dO[query c16][d = 4g..4g+3]   (k = d: lane groups 2, 3 carry zeros)
    //   bD  : B of dV  += Pd^T.dO         -> dO[queries 4g..4g+3][d = c16] (transposing read; columns d >= 8 are unused)
    //   bQ  : B of dK  += G^T.Q           -> Q [queries 4g..4g+3][d = c16]
    const pa_s16x4 dob = g < 2 ? *reinterpret_cast<const pa_s16x4*>(sD + (qb * 16 + c16) * 8 + 4 * g) : zero4;
    const int trq = (qb * 16 + 4 * g + (c16 >> 2)) * 8 + 4 * (c16 & 3);
    const pa_s16x4 bD = __builtin_amdgcn_ds_read_tr16_b64_v4i16((pa_lds_s16x4*)(sD + trq));
    const pa_s16x4 bQ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((pa_lds_s16x4*)(sQ + trq));
    // ---- sweep 1
    f32x4 P[KT];
    float m = NEG_INF;
#pragma unroll
    for (int t = 0; t < KT; ++t) {
      {
        const int kcol = t * 16 + 4 * g;
        f32x4 c;
        if (PA_FAST(t)) {
          c = pa_load4_nt(sin_p + t * TSTEP + goff);
        } else {
          const bool inrow = PA_PRED(t);
          const f32x4 ld4 = pa_load4(sin_p + (inrow ? t * TSTEP + goff : 0));
          c = inrow ? ld4 : f32x4{NEG_INF, NEG_INF, NEG_INF, NEG_INF};
        }
        if (!TILED) {
#pragma unroll
          for (int r = 0; r < 4; ++r) c[r] = (kcol + r < N) ? c[r] : NEG_INF;   // pad columns of the row are not data
        }
        P[t] = c;
        m = fmaxf(fmaxf(m, c[0]), c[1]);
        m = fmaxf(fmaxf(m, c[2]), c[3]);
      }
    }
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    if (m == NEG_INF) m = 0.f;   // rows beyond N: everything is -inf, keep the arithmetic finite
    const float mneg = -m * PA_LOG2E;   // (same exponential as the forward: exp2(S * log2(e) - m * log2(e)))
    float lsum = 0.f;
#pragma unroll
    for (int t = 0; t < KT; ++t) {
      {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = __builtin_amdgcn_exp2f(fmaf(P[t][r], PA_LOG2E, mneg));
          P[t][r] = e;
          lsum += e;
        }
      }
    }
    lsum += __shfl_xor(lsum, 16, 64);
    lsum += __shfl_xor(lsum, 32, 64);
    const float inv = lsum > 0.f ? 1.0f / lsum : 0.f;
    // dropped elements are remembered in the SIGN of P (P >= 0): |P| feeds the softmax gradient, P > 0 selects dropout(P)
    float dl = 0.f;
#pragma unroll
    for (int t = 0; t < KT; ++t) {
      {
        // A of dP^T: V[key 16t + c16][d = 4g..4g+3] (exact bf16 products, fp32 accumulation)
        const pa_s16x4 va = g < 2 ? *reinterpret_cast<const pa_s16x4*>(sV + (t * 16 + c16) * 8 + 4 * g) : zero4;
        f32x4 dp = {0.f, 0.f, 0.f, 0.f};
        dp = PA_MFMA16(va, dob, dp);
        f32x4 pr = P[t] * inv;
        if (thresh) {
          const Keep4 kw = keep4_words(seed, site, (uint64_t)(rowoff + t * 16 + 4 * g) >> 2);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const bool kp = keep4_kept(kw, r, thresh >> 16);
            dp[r] = kp ? dp[r] * dscale : 0.f;
            dl += dp[r] * pr[r];
            pr[r] = kp ? pr[r] : -pr[r];
          }
        } else {
          dl += dp[0] * pr[0] + dp[1] * pr[1] + dp[2] * pr[2] + dp[3] * pr[3];
        }
        P[t] = pr;
      }
    }
    dl += __shfl_xor(dl, 16, 64);
    dl += __shfl_xor(dl, 32, 64);
    // ---- sweep 2 (branch-free per tile;
    // dP is formed again per tile -- one 8-byte LDS read and one MFMA are cheaper than 36 more live registers)
    // G_in tiles are requested PA_LA tiles ahead of their use (a ring of PA_LA quads instead of all NT: 24 registers
    // fewer at the 168 cap -- no spills), pinned in place by scheduling barriers
    constexpr int PA_LA = 3;
    auto load_gin = [&](int t) -> f32x4 {
      if (PA_FAST(t)) {
        const f32x4 ld4 = pa_load4_nt(gin_p + (g_in_zero ? 0 : t * TSTEP + goff));
        return g_in_zero ? f32x4{0.f, 0.f, 0.f, 0.f} : ld4;
      }
      const bool inrow = PA_PRED(t) && !g_in_zero;
      const f32x4 ld4 = pa_load4(gin_p + (inrow ? t * TSTEP + goff : 0));
      return inrow ? ld4 : f32x4{0.f, 0.f, 0.f, 0.f};
    };
    f32x4 Gq[PA_LA];
#pragma unroll
    for (int t = 0; t < PA_LA && t < KT; ++t) Gq[t] = load_gin(t);
    f32x4 dq = {0.f, 0.f, 0.f, 0.f};
    const int dcol = dlane ? c16 : 0;
#pragma unroll
    for (int t = 0; t < KT; ++t) {
      const int kcol = t * 16 + 4 * g;
      const f32x4 Gin = Gq[t % PA_LA];
      if (t + PA_LA < KT) Gq[t % PA_LA] = load_gin(t + PA_LA);
      __builtin_amdgcn_sched_barrier(0);
      const pa_s16x4 va = g < 2 ? *reinterpret_cast<const pa_s16x4*>(sV + (t * 16 + c16) * 8 + 4 * g) : zero4;
      f32x4 dp = {0.f, 0.f, 0.f, 0.f};
      dp = PA_MFMA16(va, dob, dp);
      f32x4 G;
#pragma unroll
      for (int r = 0; r < 4; ++r) G[r] = fabsf(P[t][r]) * ((P[t][r] > 0.f ? dp[r] * dscale : 0.f) - dl) + Gin[r];
      if (!TILED && t * 16 + 16 > N) {   // only the last key tile has columns beyond N (uniform branch): their G must be exactly 0
#pragma unroll
        for (int r = 0; r < 4; ++r) G[r] = (kcol + r < N) ? G[r] : 0.f;
      }
      if (EDGE && !qvalid) G = f32x4{0.f, 0.f, 0.f, 0.f};
      if (PA_FAST(t)) {
        pa_store4_nt(gout_p + t * TSTEP + goff, G);
      } else if (PA_PRED(t)) {
        pa_store4_nt(gout_p + t * TSTEP + goff, G);
      }
      f32x4 Pd;
#pragma unroll
      for (int r = 0; r < 4; ++r) Pd[r] = P[t][r] > 0.f ? P[t][r] * dscale : 0.f;   // dscale == 1 without dropout
      if (EDGE && !qvalid) Pd = f32x4{0.f, 0.f, 0.f, 0.f};
      pa_s16x4 Gh, Gl, Ph, Pl;
      pa_split4(G, Gh, Gl);
      pa_split4(Pd, Ph, Pl);
      // transpose Pd and G through the wave's LDS patches: written [query][key], read [key][4 queries]
      *reinterpret_cast<pa_s16x4*>(pw + pwr) = Ph;
      *reinterpret_cast<pa_s16x4*>(pw + 256 + pwr) = Gh;
      // dQ^T += K^T . G^T : A = K[keys 16t + 4g..4g+3][d = c16 & 7] (rows d >= 8 of the result are never stored), B = G^T as it sits
      const pa_s16x4 ka = *reinterpret_cast<const pa_s16x4*>(sKT + (c16 & 7) * KSTR + t * 16 + 4 * g);
      dq = PA_MFMA16(ka, Gh, dq);
      dq = PA_MFMA16(ka, Gl, dq);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const pa_s16x4 aPh = __builtin_amdgcn_ds_read_tr16_b64_v4i16((pa_lds_s16x4*)(pw + prd));
      const pa_s16x4 aGh = __builtin_amdgcn_ds_read_tr16_b64_v4i16((pa_lds_s16x4*)(pw + 256 + prd));
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();   // (LDS operations of a wave complete in order: the low parts land after the reads above)
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      *reinterpret_cast<pa_s16x4*>(pw + pwr) = Pl;
      *reinterpret_cast<pa_s16x4*>(pw + 256 + pwr) = Gl;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const pa_s16x4 aPl = __builtin_amdgcn_ds_read_tr16_b64_v4i16((pa_lds_s16x4*)(pw + prd));
      const pa_s16x4 aGl = __builtin_amdgcn_ds_read_tr16_b64_v4i16((pa_lds_s16x4*)(pw + 256 + prd));
      // dK / dV of key tile t: accumulator column = d (lanes c16 < 8), rows = keys 16t + 4g + r; the running sums of
      // this wave live in LDS and pass through the MFMA as its C operand.
      float* accK = redw + ((wave * NT + t) * 2 + 0) * 128 + (g * 8 + dcol) * 4;
      float* accV = accK + 128;
      f32x4 dKt = *reinterpret_cast<const f32x4*>(accK), dVt = *reinterpret_cast<const f32x4*>(accV);
      dVt = PA_MFMA16(aPh, bD, dVt);
      dKt = PA_MFMA16(aGh, bQ, dKt);
      dVt = PA_MFMA16(aPl, bD, dVt);
      dKt = PA_MFMA16(aGl, bQ, dKt);
      if (dlane) {
        *reinterpret_cast<f32x4*>(accK) = dKt;
        *reinterpret_cast<f32x4*>(accV) = dVt;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();   // the next tile overwrites the patches
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    // dQ^T accumulator: column = query, rows d = 4g + r (valid for g < 2)
    if (qvalid && g < 2) {
      uint2 pk;
      pk.x = (uint32_t)f2bf(dq[0] * scale) | ((uint32_t)f2bf(dq[1] * scale) << 16);
      pk.y = (uint32_t)f2bf(dq[2] * scale) | ((uint32_t)f2bf(dq[3] * scale) << 16);
      *reinterpret_cast<uint2*>(dqkv + ((long long)b * N + qi) * D3 + h * HD + 4 * g) = pk;
    }
#undef PA_PRED
#undef PA_FAST
  };
  auto run = [&](auto kt_c) {
    for (int qb = wave; qb < nKB; qb += nwaves) {
      if (TILED && qb * 16 + 16 <= N) body(qb, std::false_type{}, kt_c);
      else body(qb, std::true_type{}, kt_c);
    }
  };
  if constexpr (RAG) pa_dispatch_kt<NT, NT>(kt, run);
  else run(std::integral_constant<int, NT>{});
  __syncthreads();
  for (int key = tid; key < N; key += blockDim.x) {
    const int t = key >> 4, kg = (key & 15) >> 2, r = key & 3;
    float a[8] = {0, 0, 0, 0, 0, 0, 0, 0}, c2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int w = 0; w < nwaves; ++w) {
      const float* bk = redw + ((w * NT + t) * 2) * 128 + kg * 32 + r;
#pragma unroll
      for (int d = 0; d < 8; ++d) {
        a[d] += bk[d * 4];
        c2[d] += bk[128 + d * 4];
      }
    }
#pragma unroll
    for (int d = 0; d < 8; ++d) a[d] *= scale;     // dK = scale * G^T.Q  (Q sits unscaled in LDS)
    bf16_t* dst = dqkv + ((long long)b * N + key) * D3 + h * HD;
    store8_bf16(dst + D, a);
    store8_bf16(dst + 2 * D, c2);
  }
}

}  // namespace mmdti
MMDTI_DEFINE_SALT_PULL(pair_attn)
using namespace mmdti;

// key tiles of 16 the MFMA kernels are instantiated for: 17 covers the reference's crop (max_atoms = 256 -> N <= 258, data/conformer.py:53,199-204)
#define PA_MAX_NT 17

static int check_common(const char* fn, int B, int N, int H, int ld) {
  MMDTI_REQUIRE(B > 0 && N > 0 && H > 0, "%s: B,N,H must be positive", fn);
  MMDTI_REQUIRE(N <= 320, "%s: N=%d exceeds the supported 320 atoms (+BOS/EOS)", fn, N);
  MMDTI_REQUIRE(ld >= N, "%s: ld (%d) < N (%d)", fn, ld, N);
  MMDTI_REQUIRE((long long)B * H <= 2147483647LL, "%s: grid too large", fn);
  return MMDTI_OK;
}

extern "C" int mmdti_pair_attn_fwd(mmdti_stream_t stream, const void* qkv_bf16, const void* bias_in, void* s_out,
                                   void* o_bf16, const unsigned char* key_pad, int B, int N, int H, int ld,
                                   float scale, float drop_p, unsigned long long seed, unsigned int site, int layout,
                                   const int* key_tiles, int rag_store) {
  const int tiled = layout & 1, compact = (layout >> 1) & 1;   // bit 0: tiled planes; bit 1: compact planes (S fp16; tiled only)
  if (int e = check_common("pair_attn_fwd", B, N, H, ld)) return e;
  MMDTI_REQUIRE((layout & ~3) == 0 && (!compact || tiled), "pair_attn_fwd: layout must be 0 (row-major fp32), 1 (tiled fp32) or 3 (tiled, fp16 logits)");
  MMDTI_REQUIRE(!key_tiles || compact, "pair_attn_fwd: key_tiles (ragged batches) needs the compact tiled pair layout (layout 3)");
  MMDTI_REQUIRE(!tiled || (ld % 4 == 0 && N <= 16 * PA_MAX_NT), "pair_attn_fwd: the tiled pair layout needs ld %% 4 == 0 and N <= 272");
  MMDTI_REQUIRE(qkv_bf16 && bias_in && s_out && o_bf16, "pair_attn_fwd: null pointer");
  MMDTI_REQUIRE(aligned16(qkv_bf16), "pair_attn_fwd: qkv must be 16-byte aligned");
  MMDTI_REQUIRE(!tiled || (aligned16(bias_in) && aligned16(s_out)), "pair_attn_fwd: tiled pair tensors must be 16-byte aligned");
  MMDTI_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "pair_attn_fwd: dropout p out of range");
  const uint32_t th = dropout_thresh(drop_p);
  const float sc = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  dim3 grid(B * H), block(256);
  hipStream_t s = (hipStream_t)stream;
  // (same eligibility rule as the backward: the two MFMA kernels share one dropout-mask generator)
  if (ld % 4 == 0 && aligned16(bias_in) && aligned16(s_out) && N <= 16 * PA_MAX_NT) {
    const int nqb = (N + 15) / 16;
    dim3 blk(nqb % 3 == 0 ? 192 : (nqb < 4 ? 64 * nqb : 256));
#define PA_M(NT, TL, FL, RG, ST)                                                                                                     \
  hipLaunchKernelGGL((pair_attn_fwd_mfma_kernel<NT, TL, FL, RG, ST>), grid, blk, 0, s, (const bf16_t*)qkv_bf16, (const ST*)bias_in, \
                     (ST*)s_out, (bf16_t*)o_bf16, key_pad, N, H, ld, scale, th, sc, (uint64_t)seed, (uint32_t)site, key_tiles, rag_store)
#define PA_MT(NT)                                                                           \
  do {                                                                                      \
    if (!tiled) PA_M(NT, false, false, false, float);                                       \
    else if (!compact) { if (nqb == NT) PA_M(NT, true, true, false, float); else PA_M(NT, true, false, false, float); } \
  } while (0)
    // The hot path (compact planes) has an instantiation for EVERY tile count: all its tiles exist, the interior ones run
    // without predicates (FULL).  In a kernel instantiated for more tiles than N has, the surplus tiles are predicated off but
    // still walked and no tile takes the fast path -- 25-45 % more time per real tile (N = 96 / 113 / 128 on the 9-tile kernel).
#define PA_MC(NT) case NT: if (key_tiles) PA_M(NT, true, true, true, _Float16); else PA_M(NT, true, true, false, _Float16); break
    if (compact) {
      switch (nqb) {
        PA_MC(1); PA_MC(2); PA_MC(3); PA_MC(4); PA_MC(5); PA_MC(6); PA_MC(7); PA_MC(8); PA_MC(9); PA_MC(10); PA_MC(11); PA_MC(12);
        PA_MC(13); PA_MC(14); PA_MC(15); PA_MC(16); PA_MC(17);
      }
    } else if (nqb <= 5) PA_MT(5); else if (nqb <= 9) PA_MT(9); else if (nqb <= 13) PA_MT(13); else PA_MT(17);
#undef PA_MC
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

extern "C" int mmdti_pair_attn_bwd(mmdti_stream_t stream, const void* qkv_bf16, const void* s, const void* do_bf16,
                                   void* g, void* dqkv_bf16, int B, int N, int H, int ld, float scale,
                                   int g_in_zero, float drop_p, unsigned long long seed, unsigned int site, int layout,
                                   const int* key_tiles) {
  // bit 0: tiled planes; bit 1: compact planes (s is fp16; tiled only); bit 2: g is bf16 (with bit 1 only; no ragged form)
  const int tiled = layout & 1, compact = (layout >> 1) & 1, g16 = (layout >> 2) & 1;
  if (int e = check_common("pair_attn_bwd", B, N, H, ld)) return e;
  MMDTI_REQUIRE((layout & ~7) == 0 && (!compact || tiled) && (!g16 || compact),
                "pair_attn_bwd: layout must be 0 (row-major fp32), 1 (tiled fp32), 3 (tiled, fp16 logits) or 7 (tiled, fp16 logits, bf16 gradients)");
  MMDTI_REQUIRE(!key_tiles || (compact && !g16), "pair_attn_bwd: key_tiles (ragged batches) needs layout 3 (tiled, fp16 logits, fp32 gradients)");
  MMDTI_REQUIRE(!tiled || (ld % 4 == 0 && N <= 16 * PA_MAX_NT), "pair_attn_bwd: the tiled pair layout needs ld %% 4 == 0 and N <= 272");
  MMDTI_REQUIRE(qkv_bf16 && s && do_bf16 && g && dqkv_bf16, "pair_attn_bwd: null pointer");
  MMDTI_REQUIRE(aligned16(qkv_bf16) && aligned16(do_bf16) && aligned16(dqkv_bf16), "pair_attn_bwd: alignment");
  MMDTI_REQUIRE(!tiled || (aligned16(s) && aligned16(g)), "pair_attn_bwd: tiled pair tensors must be 16-byte aligned");
  const uint32_t th = dropout_thresh(drop_p);
  const float sc = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  dim3 grid(B * H), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (ld % 4 == 0 && aligned16(s) && aligned16(g) && N <= 16 * PA_MAX_NT) {
    const int nqb = (N + 15) / 16;
    dim3 blk(nqb % 3 == 0 ? 192 : (nqb < 4 ? 64 * nqb : 256));
#define PA_MB(NT, TL, FL, NWV, RG, ST, GT)                                                                                     \
  hipLaunchKernelGGL((pair_attn_bwd_mfma_kernel<NT, TL, FL, NWV, RG, ST, GT>), grid, blk, 0, st, (const bf16_t*)qkv_bf16,      \
                     (const ST*)s, (const bf16_t*)do_bf16, (const GT*)g, (GT*)g, (bf16_t*)dqkv_bf16, N, H, ld, scale,           \
                     g_in_zero, th, sc, (uint64_t)seed, (uint32_t)site, key_tiles)
#define PA_MBW(NT, TL, FL, RG, ST, GT)                                                      \
  do {                                                                                      \
    if (blk.x == 192) PA_MB(NT, TL, FL, 3, RG, ST, GT); else PA_MB(NT, TL, FL, 4, RG, ST, GT); \
  } while (0)
#define PA_MBT(NT)                                                                          \
  do {                                                                                      \
    if (!tiled) PA_MBW(NT, false, false, false, float, float);                              \
    else if (!compact) { if (nqb == NT) PA_MBW(NT, true, true, false, float, float); else PA_MBW(NT, true, false, false, float, float); } \
    else { if (nqb == NT) PA_MBW(NT, true, true, false, _Float16, __bf16); else PA_MBW(NT, true, false, false, _Float16, __bf16); } \
  } while (0)
    // (compact planes with fp32 gradients -- the hot path -- have one instantiation per tile count: see the forward)
#define PA_MBC(NT)                                                                                                    \
  case NT:                                                                                                            \
    if (key_tiles) PA_MB(NT, true, true, (NT % 3 == 0 ? 3 : 4), true, _Float16, float);                               \
    else PA_MB(NT, true, true, (NT % 3 == 0 ? 3 : 4), false, _Float16, float);                                        \
    break
    if (compact && !g16) {
      switch (nqb) {
        PA_MBC(1); PA_MBC(2); PA_MBC(3); PA_MBC(4); PA_MBC(5); PA_MBC(6); PA_MBC(7); PA_MBC(8); PA_MBC(9); PA_MBC(10); PA_MBC(11);
        PA_MBC(12); PA_MBC(13); PA_MBC(14); PA_MBC(15); PA_MBC(16); PA_MBC(17);
      }
    } else if (nqb <= 5) PA_MBT(5); else if (nqb <= 9) PA_MBT(9); else if (nqb <= 13) PA_MBT(13); else PA_MBT(17);
#undef PA_MBC
#undef PA_MBW
#undef PA_MBT
#undef PA_MB
    MMDTI_LAUNCH_CHECK();
    return MMDTI_OK;
  }
#define PA_B(NCH)                                                                                                   \
  hipLaunchKernelGGL((pair_attn_bwd_kernel<NCH>), grid, block, 0, st, (const bf16_t*)qkv_bf16, (const float*)s,    \
                     (const bf16_t*)do_bf16, (float*)g, (bf16_t*)dqkv_bf16, N, H, ld, scale, g_in_zero, th, sc,            \
                     (uint64_t)seed, (uint32_t)site)
  switch ((N + 63) / 64) {
    case 1: PA_B(1); break;
    case 2: PA_B(2); break;
    case 3: PA_B(3); break;
    case 4: PA_B(4); break;
    default: PA_B(5); break;
  }
#undef PA_B
  MMDTI_LAUNCH_CHECK();
  return MMDTI_OK;
}
