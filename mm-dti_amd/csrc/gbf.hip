// Gaussian pair-distance basis (GaussianLayer.forward, models/mm_model.py:254-269; gaussian :211-224) and the
// [B,N,N,H] <-> [B,H,N,N] pair re-layout (mm_model.py:555-556), forward and backward.
//
//   y_p      = mul[e_p] * d_p + bias[e_p]                    (e_p = edge type, gathered from 961-row tables)
//   feat_pk  = exp(-0.5*((y_p - mu_k)/sigma_k)^2) / (a*sigma_k),  a = sqrt(2*3.14159), sigma = |std|+1e-5
//
// HBM-bound: per pair 4 B (dist) + 8 B (edge type, int64 as the reference collates it) in, K*2 B (bf16) out.
// A 256-thread block covers 16 pairs x 16 k-chunks of 8, so each thread's mu/sigma live in registers and every
// store is a 16-byte bf16x8.
#include "common.h"

namespace mmdti {

constexpr float GBF_A = 2.5066272160f;  // sqrt(2*3.14159) -- truncated pi exactly as mm_model.py:222-223

__global__ __launch_bounds__(256) void gbf_fwd_kernel(const float* __restrict__ dist, const long long* __restrict__ et,
                                                      const float* __restrict__ mul, const float* __restrict__ bias,
                                                      const float* __restrict__ means, const float* __restrict__ stds,
                                                      long long P, int K, int E, bf16_t* __restrict__ feat) {
  const int kc = threadIdx.x & 15;  // k chunk (8 wide); K/8 chunks looped in steps of 16
  const int pl = threadIdx.x >> 4;  // pair within the block's group of 16
  for (int k0 = kc * 8; k0 < K; k0 += 128) {  // no cross-lane traffic in this kernel: divergent trip counts are fine
    float mu[8], sg[8], cf[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      mu[i] = means[k0 + i];
      sg[i] = fabsf(stds[k0 + i]) + 1e-5f;
      cf[i] = 1.0f / (GBF_A * sg[i]);
    }
    for (long long p = (long long)blockIdx.x * 16 + pl; p < P; p += (long long)gridDim.x * 16) {
      long long e = et[p];
      e = e < 0 ? 0 : (e >= E ? E - 1 : e);
      const float y = mul[e] * dist[p] + bias[e];
      uint32_t w[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float z0 = (y - mu[2 * i]) / sg[2 * i], z1 = (y - mu[2 * i + 1]) / sg[2 * i + 1];
        float v0 = __expf(-0.5f * z0 * z0) * cf[2 * i], v1 = __expf(-0.5f * z1 * z1) * cf[2 * i + 1];
        w[i] = (uint32_t)f2bf(v0) | ((uint32_t)f2bf(v1) << 16);
      }
      *reinterpret_cast<uint4*>(feat + p * K + k0) = make_uint4(w[0], w[1], w[2], w[3]);
    }
  }
}

// Backward: per thread accumulates dmu/dsigma of its 8 k's over its pairs; the per-pair dL/dy is reduced over the
// 16 k-chunk lanes and scattered into an LDS histogram over edge types (E <= 4096), flushed with one global atomic per
// touched entry per block.
constexpr int GBF_MAXE = 4096;

__global__ __launch_bounds__(256) void gbf_bwd_kernel(const float* __restrict__ dist, const long long* __restrict__ et,
                                                      const float* __restrict__ mul, const float* __restrict__ bias,
                                                      const float* __restrict__ means, const float* __restrict__ stds,
                                                      long long P, int K, int E, const bf16_t* __restrict__ dfeat,
                                                      float* __restrict__ dmul, float* __restrict__ dbias,
                                                      float* __restrict__ dmeans, float* __restrict__ dstds) {
  extern __shared__ float hist[];  // [2][E] when E <= GBF_MAXE
  const bool use_hist = E <= GBF_MAXE;
  if (use_hist) {
    for (int i = threadIdx.x; i < 2 * E; i += 256) hist[i] = 0.f;
    __syncthreads();
  }
  const int kc = threadIdx.x & 15, pl = threadIdx.x >> 4;
  for (int kb = 0; kb < K; kb += 128) {  // uniform trip count: the body shuffles across lanes
    const int k0 = kb + kc * 8;
    const bool kact = k0 < K;
    float mu[8], sg[8], cf[8], amu[8], asg[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      mu[i] = kact ? means[k0 + i] : 0.f;
      sg[i] = kact ? fabsf(stds[k0 + i]) + 1e-5f : 1.f;
      cf[i] = 1.0f / (GBF_A * sg[i]);
      amu[i] = asg[i] = 0.f;
    }
    for (long long p0 = (long long)blockIdx.x * 16; p0 < P; p0 += (long long)gridDim.x * 16) {
      const long long p = p0 + pl;
      float dy = 0.f;
      long long e = 0;
      float d = 0.f;
      if (p < P && kact) {
        e = et[p];
        e = e < 0 ? 0 : (e >= E ? E - 1 : e);
        d = dist[p];
        const float y = mul[e] * d + bias[e];
        const uint4 u = *reinterpret_cast<const uint4*>(dfeat + p * K + k0);
        const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float dv = __uint_as_float((i & 1) ? (w[i >> 1] & 0xffff0000u) : (w[i >> 1] << 16));
          const float z = (y - mu[i]) / sg[i];
          const float val = __expf(-0.5f * z * z) * cf[i];
          const float t = dv * val;
          const float zs = z / sg[i];
          dy -= t * zs;
          amu[i] += t * zs;
          asg[i] += t * (z * zs - 1.0f / sg[i]);
        }
      }
      // reduce dy over the 16 lanes sharing this pair (lanes differ in bits 0..3)
      dy += __shfl_xor(dy, 1, 64);
      dy += __shfl_xor(dy, 2, 64);
      dy += __shfl_xor(dy, 4, 64);
      dy += __shfl_xor(dy, 8, 64);
      if (kc == 0 && p < P) {
        if (use_hist) {
          atomicAdd(&hist[e], dy * d);
          atomicAdd(&hist[E + e], dy);
        } else {
          atomicAdd(dmul + e, dy * d);
          atomicAdd(dbias + e, dy);
        }
      }
    }
    // reduce amu/asg over the 16 pair-slots of the block: lanes with equal kc are 16 apart within a wave (4 per
    // wave), then one atomic per wave per k.
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float a = amu[i], s = asg[i];
      a += __shfl_xor(a, 16, 64); a += __shfl_xor(a, 32, 64);
      s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
      if ((threadIdx.x & 63) < 16 && kact) {
        atomicAdd(dmeans + k0 + i, a);
        atomicAdd(dstds + k0 + i, stds[k0 + i] < 0.f ? -s : s);  // d|std|/dstd
      }
    }
  }
  if (use_hist) {
    __syncthreads();
    for (int i = threadIdx.x; i < E; i += 256) {
      const float a = hist[i], b = hist[E + i];
      if (a != 0.f) atomicAdd(dmul + i, a);
      if (b != 0.f) atomicAdd(dbias + i, b);
    }
  }
}

// [B,N,N,H] fp32 -> [B,H,N,ld] fp32.  One block per (b,i): the [N][H] slab is contiguous.
__global__ __launch_bounds__(256) void pair_permute_fwd_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                               int N, int H, int ld) {
  extern __shared__ float tile[];  // [N][H+1]
  const int bi = blockIdx.x, b = bi / N, i = bi - b * N;
  const float* src = x + (long long)bi * N * H;
  const int HP = H + 1;
  for (int t = threadIdx.x; t < N * H; t += 256) tile[(t / H) * HP + (t % H)] = src[t];
  __syncthreads();
  for (int t = threadIdx.x; t < H * N; t += 256) {
    const int h = t / N, j = t - h * N;
    out[(((long long)b * H + h) * N + i) * ld + j] = tile[j * HP + h];
  }
}

// [B,H,N,ld] fp32 -> [B,N,N,H] bf16
__global__ __launch_bounds__(256) void pair_permute_bwd_kernel(const float* __restrict__ g, bf16_t* __restrict__ out,
                                                               int N, int H, int ld, int tiled) {
  extern __shared__ float tile[];  // [H][NP], NP odd
  const int bi = blockIdx.x, b = bi / N, i = bi - b * N;
  const int NP = N | 1;
  for (int t = threadIdx.x; t < H * N; t += 256) {
    const int h = t / N, j = t - h * N;
    const long long off = tiled ? (long long)(b * H + h) * pair_plane(N) + pair_off(N, i, j) : (((long long)b * H + h) * N + i) * ld + j;
    tile[h * NP + j] = g[off];
  }
  __syncthreads();
  bf16_t* dst = out + (long long)bi * N * H;
  for (int t = threadIdx.x; t < N * H; t += 256) {
    const int j = t / H, h = t - j * H;
    dst[t] = f2bf(tile[h * NP + j]);
  }
}

// =====================================================================================================================
// Fused pair bias:  dist, edge type -> Gaussian basis (128) -> Linear(128,128) + GELU -> Linear(128,64) -> [B,H,N,ld]
// (mm_model.py:553-556: gbf, gbf_proj (NonLinearHead), permute) in ONE kernel.  The unfused chain writes and re-reads
// three [P,128] bf16 tensors and a [P,64] fp32 one (P = B*N*N pairs); here a wave keeps a tile of 16 pairs in
// registers from the distance to the 64 head values and both weight matrices sit in LDS.
//
// Same operand trick as attn.hip: everything is computed TRANSPOSED (lane = pair, accumulator rows = features), so the
// basis is generated directly in MFMA-operand form, and two consecutive accumulator tiles of GELU(hidden)^T are one
// K=32 operand of the second product; W2 is stored in LDS with its columns pre-permuted to that k-slot order.
// Pairs are enumerated over the padded [N][ld] plane of each molecule (q = i*ld + j): 16 consecutive q are 64
// contiguous bytes of every head plane.  Pad columns (j >= N) are written as 0.
// Optionally saves the three intermediates the unfused backward consumes (basis, pre-activation, hidden; [P,128] bf16,
// row = (b*N+i)*N+j) -- same values, same bf16 rounding points as the unfused chain.
typedef __bf16 gbf16x8 __attribute__((ext_vector_type(8)));
typedef float gf32x4 __attribute__((ext_vector_type(4)));
constexpr int GBF_K = 128, GBF_F = 128, GBF_H = 64;
constexpr int GBF_WS = 136;  // LDS row stride (elements) of the weight images: 272-B rows, conflict-free ds_read_b128 fragments

__device__ __forceinline__ gbf16x8 gbf_pack8(const float* v) {
  gbf16x8 r;
#pragma unroll
  for (int e = 0; e < 8; ++e) r[e] = (__bf16)v[e];
  return r;
}
__device__ __forceinline__ uint2 gbf_pack4(const gf32x4& a) {
  uint2 pk;
  pk.x = (uint32_t)f2bf(a[0]) | ((uint32_t)f2bf(a[1]) << 16);
  pk.y = (uint32_t)f2bf(a[2]) | ((uint32_t)f2bf(a[3]) << 16);
  return pk;
}

// Edge types as the reference collates them (int64) or narrowed on the host (int32 / int16: SURVEY 8f-3 -- 961 types fit
// 16 bits).  Branch-free so that a software-pipelined caller keeps the load in flight: one aligned 32-bit load (the low
// word of an int64 keeps sign and value of anything an embedding index can be), the int16 case picks its half.
__device__ __forceinline__ int gbf_edge(const void* et, long long base, unsigned off, int esz) {
  const char* pb = reinterpret_cast<const char*>(et) + base * esz;
  const unsigned bo = off * (unsigned)esz;
  const int w = *reinterpret_cast<const int*>(pb + (bo & ~3u));
  return esz == 2 ? (int)(short)(w >> ((bo & 2u) * 8u)) : w;
}
__device__ __forceinline__ int gbf_edge(const void* et, long long p, int esz) { return gbf_edge(et, p, 0u, esz); }

typedef float gf32x2 __attribute__((ext_vector_type(2)));
constexpr float GBF_SQ = 0.84932180028801907f;   // sqrt(log2(e) / 2): exp(-z^2 / 2) = exp2(-(GBF_SQ * z)^2)

// Eight consecutive Gaussian kernels of one pair, in MFMA B-operand form.  mu / isq / cf: the kernels' mean, GBF_SQ / sigma
// and 1 / (a sigma).  Two kernels per instruction (v_pk_add/mul_f32), one v_exp_f32 each.
__device__ __forceinline__ gbf16x8 gbf_basis8(float y, const float* mu, const float* isq, const float* cf) {
  const gf32x4 m0 = *reinterpret_cast<const gf32x4*>(mu), m1 = *reinterpret_cast<const gf32x4*>(mu + 4);
  const gf32x4 s0 = *reinterpret_cast<const gf32x4*>(isq), s1 = *reinterpret_cast<const gf32x4*>(isq + 4);
  const gf32x4 c0 = *reinterpret_cast<const gf32x4*>(cf), c1 = *reinterpret_cast<const gf32x4*>(cf + 4);
  const gf32x2 yy = {y, y};
  float v[8];
#define GBF_B2(M, S, C, LO, OUT)                                                       \
  {                                                                                    \
    const gf32x2 z = (yy - gf32x2{M[LO], M[LO + 1]}) * gf32x2{S[LO], S[LO + 1]};       \
    const gf32x2 q = -(z * z);                                                         \
    const gf32x2 e = {__builtin_amdgcn_exp2f(q[0]), __builtin_amdgcn_exp2f(q[1])};     \
    const gf32x2 r = e * gf32x2{C[LO], C[LO + 1]};                                     \
    v[OUT] = r[0]; v[OUT + 1] = r[1];                                                  \
  }
  GBF_B2(m0, s0, c0, 0, 0) GBF_B2(m0, s0, c0, 2, 2) GBF_B2(m1, s1, c1, 0, 4) GBF_B2(m1, s1, c1, 2, 6)
#undef GBF_B2
  return gbf_pack8(v);
}

// 8 waves per workgroup, two workgroups per CU (the weight images are 52 KB); grid-stride over 16-pair tiles with the
// NEXT tile's index math and loads (edge type, distance) issued before the current tile is computed.  ltab: the per-edge-
// type tables sit in LDS (E <= GBF_FWD_MAXE) -- a dependent global gather would put two round trips on every tile.
constexpr int GBF_FWD_MAXE = 4096;
// Element types of the pair planes (see pair_attn.hip): fp32, or -- compact tiled planes -- the bias as fp16 (saturating at
// 65504, -inf in the pad slots stays -inf) and the incoming gradient as bf16.
__device__ __forceinline__ void gbf_put(float* p, float v) { *p = v; }
__device__ __forceinline__ void gbf_put(_Float16* p, float v) { *p = (_Float16)fminf(v, 65504.f); }
__device__ __forceinline__ float gbf_get(const float* p) { return *p; }
__device__ __forceinline__ float gbf_get(const __bf16* p) {
  return __builtin_bit_cast(float, (uint32_t)(*reinterpret_cast<const unsigned short*>(p)) << 16);
}

template <bool SAVE, bool TILED, typename OT>
__global__ __launch_bounds__(512, 2) void gbf_bias_fwd_kernel(const float* __restrict__ dist, const void* __restrict__ et, int esz,
                                                              const float* __restrict__ mul, const float* __restrict__ bias,
                                                              const float* __restrict__ means, const float* __restrict__ stds,
                                                              const bf16_t* __restrict__ W1, const float* __restrict__ b1,
                                                              const bf16_t* __restrict__ W2, const float* __restrict__ b2,
                                                              OT* __restrict__ out, bf16_t* __restrict__ feat_out,
                                                              bf16_t* __restrict__ u_out, bf16_t* __restrict__ h_out, int B, int N,
                                                              int ld, int E, int tpm, int ugrad, int ltab,
                                                              const int* __restrict__ tile_prefix, const int* __restrict__ row_blocks) {
  static_assert(TILED || sizeof(OT) == 4, "compact pair planes exist in the tiled layout only");
  extern __shared__ __attribute__((aligned(16))) unsigned char gbf_smem[];
  bf16_t* sW1 = reinterpret_cast<bf16_t*>(gbf_smem);       // [128][136]   W1[f][k]
  bf16_t* sW2 = sW1 + GBF_F * GBF_WS;                       // [64][136]    W2[h][f], f in k-slot order
  float* sMu = reinterpret_cast<float*>(sW2 + GBF_H * GBF_WS);  // [128] means, [128] GBF_SQ/sigma, [128] 1/(a*sigma), [128] b1, [64] b2
  float* sIq = sMu + GBF_K;
  float* sCf = sIq + GBF_K;
  float* sB1 = sCf + GBF_K;
  float* sB2 = sB1 + GBF_F;
  float* sMul = sB2 + GBF_H;                                // [E], [E] when ltab
  float* sBia = sMul + E;
  const int tid = threadIdx.x, lane = tid & 63;
  for (int c = tid; c < GBF_F * (GBF_K / 8); c += 512) {
    const int row = c >> 4, col = (c & 15) * 8;
    *reinterpret_cast<uint4*>(sW1 + row * GBF_WS + col) = *reinterpret_cast<const uint4*>(W1 + row * GBF_K + col);
  }
  for (int c = tid; c < GBF_H * 32; c += 512) {  // 8-byte pieces: slot 32u+8g+4hf.. <- feature 32u+16hf+4g..
    const int row = c >> 5, pc = c & 31, u = pc >> 3, g = (pc >> 1) & 3, hf = pc & 1;
    *reinterpret_cast<uint2*>(sW2 + row * GBF_WS + 32 * u + 8 * g + 4 * hf) =
        *reinterpret_cast<const uint2*>(W2 + row * GBF_F + 32 * u + 16 * hf + 4 * g);
  }
  if (tid < GBF_K) {
    const float sg = fabsf(stds[tid]) + 1e-5f;
    sMu[tid] = means[tid];
    sIq[tid] = GBF_SQ / sg;
    sCf[tid] = 1.0f / (GBF_A * sg);
    sB1[tid] = b1[tid];
    if (tid < GBF_H) sB2[tid] = b2[tid];
  }
  if (ltab)
    for (int c = tid; c < E; c += 512) {
      sMul[c] = mul[c];
      sBia[c] = bias[c];
    }
  __syncthreads();
  const int g = lane >> 4, i = lane & 15;
  // Ragged batches (tile_prefix, tiled planes only): molecule b covers its first tile_prefix[b+1] - tile_prefix[b] tiles -- with
  // the column block slowest in a molecule's tile order these are exactly the blocks of its first key tiles; the all-padding
  // key tiles behind them are never read by the ragged pair-attention kernels and are not written.
  const bool rag = TILED && tile_prefix != nullptr;
  const int ntiles = rag ? tile_prefix[B] : B * tpm;
  const int nwaves = (int)gridDim.x * 8;
  int b_run = 0;   // (fetch is called with increasing tiles: the molecule index only moves forward)
  // Pair tiles of 16.  Row-major planes: 16 consecutive q = i*ld + j (64 contiguous bytes of every head plane).  Tiled
  // planes (the blocked rows of common.h, 16x16 tiles in MFMA accumulator order -- the layout the pair-attention kernels
  // stream): a 4x4 (query, key) block, which is again 64 contiguous bytes.  The blocks that hold a real pair cover every slot
  // of the plane; the pad keys N .. N4-1 of a real query receive -inf (the attention kernels then need no masking of pad keys).
  const int nblk = (N + 3) >> 2;
  const long long plane = TILED ? pair_plane(N) : (long long)N * ld;
  // flags: bit 0 valid pair, bit 1 slot inside the plane, bit 2 whole block past N
  auto fetch = [&](int tile, int& b, int& q, int& flags, unsigned& pl, int& e, float& d) {
    tile = __builtin_amdgcn_readfirstlane(tile < ntiles ? tile : ntiles - 1);
    int tq;
    if (rag) {
      while (b_run + 1 < B && tile >= tile_prefix[b_run + 1]) ++b_run;
      b = b_run;
      tq = tile - tile_prefix[b];
    } else {
      b = (int)((unsigned)tile / (unsigned)tpm);
      tq = tile - b * tpm;
    }
    int ii, jj;
    bool inplane = true, past = false;
    if (TILED) {
      // (query block fastest: inside a 16x16 tile the 64-byte segments of rb & 3 = 0..3 are consecutive in memory, so the
      //  eight waves of a workgroup write -- and the backward reads -- whole 128-byte lines together)
      // packed token rows (row_blocks): molecule b's queries stop at its representative pad row -- only its first row_blocks[b]
      // 4-row blocks are produced (the pair-attention kernels read no query row past it)
      const int rbk = (rag && row_blocks) ? row_blocks[b] : nblk;
      const int cb = (int)((unsigned)tq / (unsigned)rbk), rb = tq - cb * rbk;
      past = 4 * rb >= N || 4 * cb >= N;   // (the same for every lane of the wave; no such block is enumerated: nothing is stored for it)
      ii = 4 * rb + (i >> 2);
      jj = 4 * cb + (i & 3);
      inplane = ii < N;                    // (a row past N has no slot)
      q = (int)pair_off(N, inplane ? ii : 0, jj);
    } else {
      q = tq * 16 + i;
      ii = (int)((unsigned)q / (unsigned)ld);
      jj = q - ii * ld;
      inplane = q < plane;
    }
    const bool valid = ii < N && jj < N;
    flags = (valid ? 1 : 0) | (inplane ? 2 : 0) | (past ? 4 : 0);
    pl = (unsigned)((valid ? ii : 0) * N + (valid ? jj : 0));
    const long long mol = (long long)b * N * N;
    const int e32 = gbf_edge(et, mol, pl, esz);
    e = e32 < 0 ? 0 : (e32 >= E ? E - 1 : e32);
    d = (dist + mol)[pl];
  };
  int b, q, flags, e, n_b, n_q, n_flags, n_e;
  unsigned pl, n_pl;
  float d, n_d;
  int tile = (int)blockIdx.x * 8 + (tid >> 6);
  fetch(tile, b, q, flags, pl, e, d);
  for (; tile < ntiles; tile += nwaves) {
    fetch(tile + nwaves, n_b, n_q, n_flags, n_pl, n_e, n_d);
    const bool valid = flags & 1, inplane = flags & 2, past = flags & 4;
    const float padv = TILED ? -INFINITY : 0.f;
    OT* ob = out + (long long)b * GBF_H * plane + q;
    if (TILED && past) {
    } else {
    const long long p = (long long)b * N * N + pl;
    const float y = ltab ? sMul[e] * d + sBia[e] : mul[e] * d + bias[e];
    // Gaussian basis in B-operand form: lane (g, pair i) holds k = 32c + 8g + 0..7
    gbf16x8 fB[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int k0 = 32 * c + 8 * g;
      fB[c] = gbf_basis8(y, sMu + k0, sIq + k0, sCf + k0);
      if (SAVE && valid) *reinterpret_cast<gbf16x8*>(feat_out + p * GBF_K + k0) = fB[c];
    }
    // hidden^T = W1 . basis^T (+ b1), GELU; pairs of accumulator tiles become the K=32 operands of the second product
    gbf16x8 hB[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      gf32x4 hv[2];
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const int ft = 2 * u + hf;
        gf32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const gbf16x8 wf = *reinterpret_cast<const gbf16x8*>(sW1 + (16 * ft + i) * GBF_WS + 32 * c + 8 * g);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, fB[c], acc, 0, 0, 0);
        }
        const gf32x4 bb = *reinterpret_cast<const gf32x4*>(sB1 + 16 * ft + 4 * g);
        acc += bb;
        if (SAVE && ugrad) {   // the saved tensor holds gelu'(pre-activation): the erf and the Gaussian are in hand here, and the
                               // backward kernel -- alone on the chip at the very end of the step -- is left with a multiply
          f32x2_t y0, g0, y1, g1;
          gelu_erf_both2(f32x2_t{acc[0], acc[1]}, y0, g0);
          gelu_erf_both2(f32x2_t{acc[2], acc[3]}, y1, g1);
          acc = gf32x4{y0[0], y0[1], y1[0], y1[1]};
          const gf32x4 gq = {g0[0], g0[1], g1[0], g1[1]};
          if (valid) *reinterpret_cast<uint2*>(u_out + p * GBF_F + 16 * ft + 4 * g) = gbf_pack4(gq);
        } else {
          if (SAVE && valid) *reinterpret_cast<uint2*>(u_out + p * GBF_F + 16 * ft + 4 * g) = gbf_pack4(acc);
          const f32x2_t y0 = gelu_erf2(f32x2_t{acc[0], acc[1]}), y1 = gelu_erf2(f32x2_t{acc[2], acc[3]});
          acc = gf32x4{y0[0], y0[1], y1[0], y1[1]};
        }
        if (SAVE && valid) *reinterpret_cast<uint2*>(h_out + p * GBF_F + 16 * ft + 4 * g) = gbf_pack4(acc);
        hv[hf] = acc;
        __builtin_amdgcn_sched_barrier(0);  // keep the unrolled tiles in order: hoisted weight fragments would eat the register file
      }
      float v[8] = {hv[0][0], hv[0][1], hv[0][2], hv[0][3], hv[1][0], hv[1][1], hv[1][2], hv[1][3]};
      hB[u] = gbf_pack8(v);
    }
    // bias^T = W2 . hidden^T (+ b2): accumulator rows = heads 16*ht + 4g + r, column = pair
#pragma unroll
    for (int ht = 0; ht < 4; ++ht) {
      gf32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const gbf16x8 wf = *reinterpret_cast<const gbf16x8*>(sW2 + (16 * ht + i) * GBF_WS + 32 * u + 8 * g);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, hB[u], acc, 0, 0, 0);
      }
      const gf32x4 bb = *reinterpret_cast<const gf32x4*>(sB2 + 16 * ht + 4 * g);
      if (inplane) {
#pragma unroll
        for (int r = 0; r < 4; ++r) gbf_put(ob + (long long)(16 * ht + 4 * g + r) * plane, valid ? acc[r] + bb[r] : padv);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    }
    b = n_b; q = n_q; flags = n_flags; pl = n_pl; e = n_e; d = n_d;
  }
}

// =====================================================================================================================
// Fused pair-bias backward, the per-pair half (mm_model.py:553-556 backward):
//   dO = bf16(G)                                  [P,64]   -> written (A operand of dW2 = dO^T.hidden, and db2)
//   du = bf16((dO.W2) * gelu'(u))                 [P,128]  -> written (A operand of dW1 = du^T.basis, and db1)
//   dbasis = bf16(du.W1) -> Gaussian backward -> d mul/bias (LDS histogram over edge types), d means/stds
// in ONE pass over G: replaces the G re-layout, two dX GEMMs over P = 4.3 M rows and the Gaussian-backward kernel, and
// the [P,128] dbasis tensor never exists.  Same transposed-operand scheme and pair enumeration as the forward kernel;
// the two weight-gradient GEMMs (contraction over the 4.3 M pairs) stay on the GEMM kernel.
constexpr int GBF_W2S = 72;   // LDS row stride of W2^T [128 f][64 h]: 144-B rows, conflict-free ds_read_b128 fragments

template <bool TILED, typename GT>
__global__ __launch_bounds__(256, 2) void gbf_bias_bwd_kernel(const GT* __restrict__ gsrc, const float* __restrict__ dist,
                                                           const void* __restrict__ et, int esz, const float* __restrict__ mul,
                                                           const float* __restrict__ bias, const float* __restrict__ means,
                                                           const float* __restrict__ stds, const bf16_t* __restrict__ W1,
                                                           const bf16_t* __restrict__ W2, const bf16_t* __restrict__ u_in,
                                                           bf16_t* __restrict__ do_out, bf16_t* __restrict__ du_out,
                                                           float* __restrict__ dmul, float* __restrict__ dbias,
                                                           float* __restrict__ dmeans, float* __restrict__ dstds, int B, int N, int ld,
                                                           int E, int tpm, int ugrad) {
  extern __shared__ __attribute__((aligned(16))) unsigned char gbf_smem[];
  bf16_t* sW1T = reinterpret_cast<bf16_t*>(gbf_smem);   // [128 k][136]  W1^T[k][f], f in k-slot order
  bf16_t* sW2T = sW1T + GBF_K * GBF_WS;                  // [128 f][72]   W2^T[f][h]
  float* sMu = reinterpret_cast<float*>(sW2T + GBF_F * GBF_W2S);
  float* sIs = sMu + GBF_K;
  float* sCf = sIs + GBF_K;
  float* hist = sCf + GBF_K;                             // [2][E]
  float* sMul = hist + 2 * E;                            // [E] mul, [E] bias (the per-pair affine of the Gaussian layer)
  float* sBia = sMul + E;
  const int tid = threadIdx.x, lane = tid & 63;
  for (int c = tid; c < E; c += 256) {
    sMul[c] = mul[c];
    sBia[c] = bias[c];
  }
  // W1^T with the feature (contraction) axis in k-slot order: slot 32u+8g+4hf+e <- feature 32u+16hf+4g+e
  for (int c = tid; c < GBF_K * GBF_F; c += 256) {
    const int f = c >> 7, k = c & 127;                   // coalesced read of W1[f][k]
    const int u = f >> 5, hf = (f >> 4) & 1, g = (f >> 2) & 3, e = f & 3;
    sW1T[k * GBF_WS + 32 * u + 8 * g + 4 * hf + e] = W1[c];
  }
  for (int c = tid; c < GBF_H * GBF_F; c += 256) {
    const int h = c >> 7, f = c & 127;
    sW2T[f * GBF_W2S + h] = W2[c];
  }
  if (tid < GBF_K) {
    const float sg = fabsf(stds[tid]) + 1e-5f;
    sMu[tid] = means[tid];
    sIs[tid] = 1.0f / sg;
    sCf[tid] = 1.0f / (GBF_A * sg);
  }
  for (int c = tid; c < 2 * E; c += 256) hist[c] = 0.f;
  __syncthreads();
  const int g = lane >> 4, i = lane & 15;
  const long long ntiles = (long long)B * tpm;
  const long long nwaves = (long long)gridDim.x * 4;
  const int nblk = (N + 3) >> 2;
  const long long plane = TILED ? pair_plane(N) : (long long)N * ld;
  // d means / d stds partial sums of this lane: k = 16*kt + 4g + r
  gf32x4 amu[8], asg[8];
#pragma unroll
  for (int kt = 0; kt < 8; ++kt) amu[kt] = asg[kt] = gf32x4{0.f, 0.f, 0.f, 0.f};
  // Software pipeline over pair tiles: the index math and every global load of tile t+1 (edge type, distance, the 16 head
  // values of G, the 8 saved pre-activation quads) are issued before tile t is computed -- at 2 waves per SIMD nothing
  // else hides the ~2 us round trips.
  auto fetch = [&](long long tile, bool& act, bool& valid, long long& p, int& e, float& d, float (&gv)[16], uint2 (&up)[8]) {
    act = false;
    valid = false;
    p = 0;
    e = 0;
    d = 0.f;
    if (tile >= ntiles) return;
    const int b = (int)(tile / tpm);
    const int tq = (int)(tile - (long long)b * tpm);
    int q, ii, jj;
    if (TILED) {
      const int rb = tq / nblk, cb = tq - rb * nblk;
      if (4 * rb >= N || 4 * cb >= N) return;            // whole block past N (the same for every lane of the wave)
      ii = 4 * rb + (i >> 2);
      jj = 4 * cb + (i & 3);
      q = (int)pair_off(N, ii < N ? ii : 0, jj);
    } else {
      q = tq * 16 + i;
      ii = q / ld;
      jj = q - ii * ld;
    }
    act = true;
    valid = ii < N && jj < N;
    p = ((long long)b * N + (valid ? ii : 0)) * N + (valid ? jj : 0);
    const int e32 = gbf_edge(et, p, esz);
    e = e32 < 0 ? 0 : (e32 >= E ? E - 1 : e32);
    d = dist[p];
    const GT* gp = gsrc + (long long)b * GBF_H * plane + (valid ? q : 0);
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int j = 0; j < 8; ++j) gv[c * 8 + j] = valid ? gbf_get(gp + (long long)(32 * c + 8 * g + j) * plane) : 0.f;
#pragma unroll
    for (int ft = 0; ft < 8; ++ft) {
      up[ft] = make_uint2(0u, 0u);
      if (valid) up[ft] = *reinterpret_cast<const uint2*>(u_in + p * GBF_F + 16 * ft + 4 * g);
    }
  };
  bool act, valid, n_act, n_valid;
  long long p, n_p;
  int e, n_e;
  float d, n_d, gv[16], n_gv[16];
  uint2 upre[8], n_up[8];
  long long tile = (long long)blockIdx.x * 4 + (tid >> 6);
  fetch(tile, act, valid, p, e, d, gv, upre);
  for (; tile < ntiles; tile += nwaves) {
    fetch(tile + nwaves, n_act, n_valid, n_p, n_e, n_d, n_gv, n_up);
    if (act) {
    const float y = sMul[e] * d + sBia[e];
    // dO^T in B-operand form: lane (g, pair i) holds heads 32c + 8g + 0..7
    gbf16x8 oB[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      oB[c] = gbf_pack8(gv + 8 * c);
      if (valid) *reinterpret_cast<gbf16x8*>(do_out + p * GBF_H + 32 * c + 8 * g) = oB[c];
    }
    // du^T = (W2^T . dO^T) * gelu'(u): accumulator rows = features 16*ft + 4g + r
    gbf16x8 uB[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      gf32x4 dv[2];
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const int ft = 2 * u + hf;
        gf32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const gbf16x8 wf = *reinterpret_cast<const gbf16x8*>(sW2T + (16 * ft + i) * GBF_W2S + 32 * c + 8 * g);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, oB[c], acc, 0, 0, 0);
        }
        const uint2 up = upre[ft];
        if (ugrad) {   // (the forward saved gelu' itself)
          acc[0] *= __uint_as_float(up.x << 16);
          acc[1] *= __uint_as_float(up.x & 0xffff0000u);
          acc[2] *= __uint_as_float(up.y << 16);
          acc[3] *= __uint_as_float(up.y & 0xffff0000u);
        } else {
          acc[0] *= gelu_erf_grad(__uint_as_float(up.x << 16));
          acc[1] *= gelu_erf_grad(__uint_as_float(up.x & 0xffff0000u));
          acc[2] *= gelu_erf_grad(__uint_as_float(up.y << 16));
          acc[3] *= gelu_erf_grad(__uint_as_float(up.y & 0xffff0000u));
        }
        if (valid) *reinterpret_cast<uint2*>(du_out + p * GBF_F + 16 * ft + 4 * g) = gbf_pack4(acc);
        dv[hf] = acc;
        __builtin_amdgcn_sched_barrier(0);
      }
      float v[8] = {dv[0][0], dv[0][1], dv[0][2], dv[0][3], dv[1][0], dv[1][1], dv[1][2], dv[1][3]};
      uB[u] = gbf_pack8(v);
    }
    // dbasis^T = W1^T . du^T: rows = Gaussian kernels 16*kt + 4g + r; Gaussian backward on the accumulator layout
    float dy = 0.f;
#pragma unroll
    for (int kt = 0; kt < 8; ++kt) {
      gf32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const gbf16x8 wf = *reinterpret_cast<const gbf16x8*>(sW1T + (16 * kt + i) * GBF_WS + 32 * u + 8 * g);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, uB[u], acc, 0, 0, 0);
      }
      const gf32x4 mu = *reinterpret_cast<const gf32x4*>(sMu + 16 * kt + 4 * g);
      const gf32x4 is = *reinterpret_cast<const gf32x4*>(sIs + 16 * kt + 4 * g);
      const gf32x4 cf = *reinterpret_cast<const gf32x4*>(sCf + 16 * kt + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float dvr = __uint_as_float(((uint32_t)f2bf(acc[r])) << 16);   // dbasis passes through bf16 like the unfused chain
        const float z = (y - mu[r]) * is[r];
        const float val = __expf(-0.5f * z * z) * cf[r];
        const float t = valid ? dvr * val : 0.f;
        const float zs = z * is[r];
        dy -= t * zs;
        amu[kt][r] += t * zs;
        asg[kt][r] += t * (z * zs - is[r]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    dy += __shfl_xor(dy, 16, 64);
    dy += __shfl_xor(dy, 32, 64);
    if (g == 0 && valid) {
      atomicAdd(&hist[e], dy * d);
      atomicAdd(&hist[E + e], dy);
    }
    }  // act
    act = n_act; valid = n_valid; p = n_p; e = n_e; d = n_d;
#pragma unroll
    for (int j = 0; j < 16; ++j) gv[j] = n_gv[j];
#pragma unroll
    for (int j = 0; j < 8; ++j) upre[j] = n_up[j];
  }
  // d means / d stds: sum over the 16 pair lanes of each lane group, one atomic per (wave, k)
#pragma unroll
  for (int kt = 0; kt < 8; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float a = amu[kt][r], sgv = asg[kt][r];
      a += __shfl_xor(a, 1, 64); a += __shfl_xor(a, 2, 64); a += __shfl_xor(a, 4, 64); a += __shfl_xor(a, 8, 64);
      sgv += __shfl_xor(sgv, 1, 64); sgv += __shfl_xor(sgv, 2, 64); sgv += __shfl_xor(sgv, 4, 64); sgv += __shfl_xor(sgv, 8, 64);
      if (i == 0) {
        const int k = 16 * kt + 4 * g + r;
        atomicAdd(dmeans + k, a);
        atomicAdd(dstds + k, stds[k] < 0.f ? -sgv : sgv);   // d|std|/dstd
      }
    }
  __syncthreads();
  for (int c = tid; c < E; c += 256) {
    const float a = hist[c], bb = hist[E + c];
    if (a != 0.f) atomicAdd(dmul + c, a);
    if (bb != 0.f) atomicAdd(dbias + c, bb);
  }
}

// =====================================================================================================================
// Fused pair-bias backward, COMPLETE: everything mm_model.py:553-556 owes its parameters, from G = dL/d(bias) and the
// batch's distances / edge types alone.  The forward saves NOTHING (its three [P,128] bf16 intermediates were 768 of its
// 1 036 bytes per atom pair, and the two weight-gradient GEMMs then read 3.9 GB more): the Gaussian basis, the hidden
// pre-activation and GELU are recomputed here (cheaper than the 1.5 KB per pair they replace), and the weight gradients,
// whose contraction runs over ALL pairs, are accumulated in registers by the same workgroups.
//
// One persistent workgroup of 8 waves per CU.  Iteration = 128 pairs (a 16-pair tile per wave, same tile enumeration and
// transposed-operand scheme as the forward kernel):
//   phase A (wave = its 16 pairs):  y -> basis^T (B operand) -> rows of LDS tile T0;  dO^T = bf16(G);  per 16 hidden
//                        features: du_raw^T = W2^T.dO^T, u^T = W1.basis^T + b1, (h, gelu') = GELU(u),
//                        du^T = du_raw^T * gelu' -> rows of LDS tile T1; h stays packed in registers
//   phase B1 (wave w = feature / kernel block w, all 128 pairs):
//                        dW1[16w.., :] += du^T x basis and db1 (contraction over the pairs: transposing reads of T1 / T0);
//                        dbasis^T[16w.., pairs] = W1^T.du^T (A by transposing reads of the W1 image, B = rows of T1), then
//                        the Gaussian backward on the accumulator: d means / d stds of the wave's 16 kernels stay in 8
//                        registers, the per-pair dy partials meet in an LDS row
//   phase B2:            T0 / T1 refilled with h and dO;  dW2[:, 16w..] += dO^T x h, db2;  dy -> per-edge-type histogram.
// The [pair][feature] tiles have 288-byte rows with the feature index XOR-ed by 64 on odd 8-row groups: writes are 8/16
// bytes per lane, the transposing reads of a half-wave hit 32 distinct bank pairs.
constexpr int GBF_LW = 144;       // LDS row stride (elements) of W1 [128 f][144] and of the two pair tiles [128 pairs][144]
constexpr int GBF_FULL_MAXE = 1536;   // 4 per-edge-type fp32 tables next to 126 KB of tiles in the CU's 160 KB
// per-workgroup slab of partial gradients (fp32 elements): dW1 [F][K] | dW2 [H][F] | db1 [F] | db2 [H] | dmeans [K] | dstds [K] | dmul [E] | dbias [E]
constexpr int GBF_SLAB_B1 = GBF_F * GBF_K + GBF_H * GBF_F, GBF_SLAB_B2 = GBF_SLAB_B1 + GBF_F, GBF_SLAB_MU = GBF_SLAB_B2 + GBF_H,
              GBF_SLAB_SG = GBF_SLAB_MU + GBF_K, GBF_SLAB = GBF_SLAB_SG + GBF_K;
constexpr size_t gbf_full_smem(int E) {
  return (size_t)(GBF_F * GBF_LW + GBF_F * GBF_W2S + 2 * 128 * GBF_LW) * 2 + (size_t)(4 * GBF_K + GBF_F + 2 * 128 + 4 * E) * 4;
}

typedef short gs16x4 __attribute__((ext_vector_type(4)));
typedef short gs16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) gs16x4 glds_s16x4;
__device__ __forceinline__ gbf16x8 gbf_tr8(const bf16_t* a0, const bf16_t* a1) {
  const gs16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((glds_s16x4*)a0);
  const gs16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((glds_s16x4*)a1);
  const gs16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(gbf16x8, v);
}

// TILED: a tile is one 4x4 block of pairs, tpm = ceil(N/4)^2 per molecule (only blocks that hold a real pair are enumerated).
template <bool TILED, typename GT>
__global__ __launch_bounds__(512, 1) void gbf_bias_bwd_full_kernel(
    const GT* __restrict__ gsrc, const float* __restrict__ dist, const void* __restrict__ et, int esz, const float* __restrict__ mul,
    const float* __restrict__ bias, const float* __restrict__ means, const float* __restrict__ stds, const bf16_t* __restrict__ W1,
    const float* __restrict__ b1, const bf16_t* __restrict__ W2, float* __restrict__ dW1, float* __restrict__ db1, float* __restrict__ dW2,
    float* __restrict__ db2, float* __restrict__ dmul, float* __restrict__ dbias, float* __restrict__ dmeans, float* __restrict__ dstds, int B,
    int N, int ld, int E, int tpm, const int* __restrict__ tile_prefix, const int* __restrict__ row_blocks, float* __restrict__ slab) {
  extern __shared__ __attribute__((aligned(16))) unsigned char gbf_smem[];
  bf16_t* sW1 = reinterpret_cast<bf16_t*>(gbf_smem);            // [128 f][144]  W1[f][k]
  bf16_t* sW2T = sW1 + GBF_F * GBF_LW;                           // [128 f][72]   W2^T[f][h]
  bf16_t* T0 = sW2T + GBF_F * GBF_W2S;                           // [128 pairs][144]: basis, later hidden
  bf16_t* T1 = T0 + 128 * GBF_LW;                                // [128 pairs][144]: du, later dO
  float* sMu = reinterpret_cast<float*>(T1 + 128 * GBF_LW);
  float* sIs = sMu + GBF_K;
  float* sCf = sIs + GBF_K;
  float* sIq = sCf + GBF_K;                                      // GBF_SQ / sigma (phase A's exp2 form)
  float* sB1 = sIq + GBF_K;
  float* sY = sB1 + GBF_F;                                       // [128] y of the iteration's pairs
  float* sDy = sY + 128;                                         // [128] dL/dy partial sums
  float* hist = sDy + 128;                                       // [2][E]
  float* sMul = hist + 2 * E;                                    // [E]
  float* sBia = sMul + E;                                        // [E]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int c = tid; c < E; c += 512) {
    sMul[c] = mul[c];
    sBia[c] = bias[c];
  }
  for (int c = tid; c < GBF_F * (GBF_K / 8); c += 512) {
    const int row = c >> 4, col = (c & 15) * 8;
    *reinterpret_cast<uint4*>(sW1 + row * GBF_LW + col) = *reinterpret_cast<const uint4*>(W1 + row * GBF_K + col);
  }
  for (int c = tid; c < GBF_H * GBF_F; c += 512) {
    const int h = c >> 7, f = c & 127;
    sW2T[f * GBF_W2S + h] = W2[c];
  }
  if (tid < GBF_K) {
    const float sg = fabsf(stds[tid]) + 1e-5f;
    sMu[tid] = means[tid];
    sIs[tid] = 1.0f / sg;
    sCf[tid] = 1.0f / (GBF_A * sg);
    sIq[tid] = GBF_SQ / sg;
    sB1[tid] = b1[tid];
    sDy[tid] = 0.f;
  }
  for (int c = tid; c < 2 * E; c += 512) hist[c] = 0.f;
  __syncthreads();
  const int g = lane >> 4, i = lane & 15;
  const int ploc = 16 * wave + i;                                 // this lane's row in the pair tiles
  const int pswz = ((ploc >> 3) & 1) << 6;
  // Swizzled column c ^ swz (swz = 0 or 64) = (c < 64 ? c + swz : c - 64 + (64 ^ swz)): two bases, compile-time offsets --
  // the compiler would otherwise hoist one address register per access out of the loop and spill them.
#define GBF_SWZ(lo, hi, c) ((c) < 64 ? (lo) + (c) : (hi) + ((c) - 64))
  bf16_t* const r0lo = T0 + ploc * GBF_LW + pswz;               // this lane's pair row of T0 / T1
  bf16_t* const r0hi = T0 + ploc * GBF_LW + (64 ^ pswz);
  bf16_t* const r1lo = r0lo + 128 * GBF_LW;                      // (T1 = T0 + 128 rows: the same registers, immediate offsets)
  bf16_t* const r1hi = r0hi + 128 * GBF_LW;
  const int tswz = (g & 1) << 6;                                 // transposing reads: pair rows 32s + 8g + (i>>2) [+4]
  const int trow = (8 * g + (i >> 2)) * GBF_LW + 4 * (i & 3);
  const bf16_t* const t0lo = T0 + trow + tswz;
  const bf16_t* const t0hi = T0 + trow + (64 ^ tswz);
  const bf16_t* const t1lo = t0lo + 128 * GBF_LW;
  const bf16_t* const t1hi = t0hi + 128 * GBF_LW;               // (phase B2: the low parts of dO in the unused half of T1's columns)
  const bf16_t* const t0w = T0 + trow + ((16 * wave) ^ tswz);   // column block = this wave
  const bf16_t* const t1w = t0w + 128 * GBF_LW;
  const int rswz = ((i >> 3) & 1) << 6;                          // row reads of T1: pair row 16j + i
  const bf16_t* const d1lo = T1 + i * GBF_LW + 8 * g + rswz;
  const bf16_t* const d1hi = T1 + i * GBF_LW + 8 * g + (64 ^ rswz);
  // byte offsets (from the start of the LDS block) of the loop-invariant images; see the laundering note in the loop
  unsigned o_w1row = (unsigned)(i * GBF_LW + 8 * g) * 2;                                   // A fragments of W1 (rows 16ft + i)
  unsigned o_w2row = (unsigned)(GBF_F * GBF_LW + i * GBF_W2S + 8 * g) * 2;                 // ... and of W2^T
  unsigned o_w1tr = (unsigned)((8 * g + (i >> 2)) * GBF_LW + 16 * wave + 4 * (i & 3)) * 2; // W1^T fragment, k block = wave
  const unsigned o_f32 = (unsigned)(GBF_F * GBF_LW + GBF_F * GBF_W2S + 2 * 128 * GBF_LW) * 2;  // sMu
  unsigned o_gconst = o_f32 + (unsigned)(8 * g) * 4;             // Gaussian constants of kernels 32c + 8g + 0..7 (phase A)
  unsigned o_b1row = o_f32 + (unsigned)(4 * GBF_K + 4 * g) * 4;
  unsigned o_kconst = o_f32 + (unsigned)(16 * wave + 4 * g) * 4; // phase B1's Gaussian constants: kernels k = 16*wave + 4g + r
  // (ragged batches: tile_prefix as in the forward kernel -- here a molecule's tiles are its 4x4 blocks of REAL pairs, column
  //  block slowest, so a prefix of them covers its first key tiles; the gradient is zero, and never written, behind them)
  const bool rag = TILED && tile_prefix != nullptr;
  const int ntiles = rag ? tile_prefix[B] : B * tpm;
  int b_run = 0;
  const int nb = (N + 3) >> 2;
  const long long plane = TILED ? pair_plane(N) : (long long)N * ld;
  gf32x4 amu = {0.f, 0.f, 0.f, 0.f}, asg = {0.f, 0.f, 0.f, 0.f};
  gf32x4 aW1[8], aW2[4], aB1 = {0.f, 0.f, 0.f, 0.f}, aB2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < 8; ++t) aW1[t] = gf32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < 4; ++t) aW2[t] = gf32x4{0.f, 0.f, 0.f, 0.f};
  const gs16x8 o16 = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
  const gbf16x8 ones = __builtin_bit_cast(gbf16x8, o16);

  // Branch-free: a tile past the end is clamped to the last one and flagged inactive, a pad lane reads pair (0, 0) of its
  // molecule -- so every load of the NEXT iteration is in flight, unconditionally, while this one computes.
  auto fetch = [&](int tile, bool& act, bool& valid, float& d, int& e, float (&gv)[16]) {
    tile = __builtin_amdgcn_readfirstlane(tile);          // (wave-uniform: tile = 8 * workgroup + wave)
    act = tile < ntiles;
    tile = act ? tile : ntiles - 1;
    int b, tq;
    if (rag) {
      while (b_run + 1 < B && tile >= tile_prefix[b_run + 1]) ++b_run;
      b = b_run;
      tq = tile - tile_prefix[b];
    } else {
      b = (int)((unsigned)tile / (unsigned)tpm);
      tq = tile - b * tpm;
    }
    int q, ii, jj;
    if (TILED) {
      const int rbk = (rag && row_blocks) ? row_blocks[b] : nb;               // (packed token rows: see the forward kernel)
      const int cb = (int)((unsigned)tq / (unsigned)rbk), rb = tq - cb * rbk;   // query block fastest: see the forward kernel
      ii = 4 * rb + (i >> 2);
      jj = 4 * cb + (i & 3);
      q = (int)pair_off(N, ii < N ? ii : 0, jj);
    } else {
      q = tq * 16 + i;
      ii = (int)((unsigned)q / (unsigned)ld);
      jj = q - ii * ld;
    }
    valid = act && ii < N && jj < N;
    // scalar base of the molecule + 32-bit lane offsets (host: N*N and 64*plane fit 31 bits)
    const long long mol = (long long)b * N * N;
    const unsigned pl = (unsigned)((valid ? ii : 0) * N + (valid ? jj : 0));
    const int e32 = gbf_edge(et, mol, pl, esz);
    e = e32 < 0 ? 0 : (e32 >= E ? E - 1 : e32);
    d = (dist + mol)[pl];
    const GT* mb = gsrc + (long long)b * GBF_H * plane;
    const unsigned voff = (unsigned)(8 * g) * (unsigned)plane + (unsigned)(valid ? q : 0);
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int j = 0; j < 8; ++j) gv[c * 8 + j] = gbf_get(mb + (long long)(32 * c + j) * plane + voff);
  };

  bool act, valid, n_act, n_valid;
  float d, n_d, gv[16], n_gv[16];
  int e, n_e;
  const int stride = (int)gridDim.x * 8;
  int tile = (int)blockIdx.x * 8 + wave;
  fetch(tile, act, valid, d, e, gv);
  // (every wave of the workgroup runs the same number of iterations: the loop bound is on the workgroup's first tile)
  for (int base = (int)blockIdx.x * 8; base < ntiles; base += stride, tile += stride) {
    // The weight / constant images in LDS never change inside the loop, and the compiler knows: left alone it hoists some
    // 300 registers' worth of their reads out of the loop and spills them.  Laundering the (integer) offsets pins the
    // reads where they are written and keeps the pointers in the LDS address space.
    asm volatile("" : "+v"(o_w1row), "+v"(o_w2row), "+v"(o_w1tr), "+v"(o_gconst), "+v"(o_b1row), "+v"(o_kconst));
    const bf16_t* w1row = reinterpret_cast<const bf16_t*>(gbf_smem + o_w1row);
    const bf16_t* w2row = reinterpret_cast<const bf16_t*>(gbf_smem + o_w2row);
    const bf16_t* w1tr = reinterpret_cast<const bf16_t*>(gbf_smem + o_w1tr);
    const float* gconst = reinterpret_cast<const float*>(gbf_smem + o_gconst);
    const float* b1row = reinterpret_cast<const float*>(gbf_smem + o_b1row);
    const float* kconst = reinterpret_cast<const float*>(gbf_smem + o_kconst);
    fetch(tile + stride, n_act, n_valid, n_d, n_e, n_gv);
    // ------------------------------------------------------------------------------------------------ phase A
    // The incoming gradient enters the two products that contract it -- du = (W2^T . dO) . gelu' here, dW2 / db2 in phase B2 -- as a
    // bf16 high part + a bf16 low part (16 mantissa bits, two MFMAs).  Rows of G sum to zero over the keys, and every (pad query,
    // real key) pair of a molecule has the same basis and hidden vector: in exact arithmetic their contributions cancel, with G
    // rounded to bf16 they left 2^-9-sized residues summed coherently over thousands of pairs (round 4: the four table gradients of
    // a mixed-length batch 5 x further from the fp32 oracle than the 16-bit contract's own emulation is).
    constexpr bool SPLIT_G = sizeof(GT) == 4;      // (a bf16 gradient chain -- opt-in -- has no low part)
    gbf16x8 oB[2], oL[2], hB[4];
    if (!valid) {
#pragma unroll
      for (int j = 0; j < 16; ++j) gv[j] = 0.f;
    }
    oB[0] = gbf_pack8(gv);
    oB[1] = gbf_pack8(gv + 8);
    if constexpr (SPLIT_G) {
      float lo[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) lo[j] = gv[j] - (float)oB[j >> 3][j & 7];
      oL[0] = gbf_pack8(lo);
      oL[1] = gbf_pack8(lo + 8);
    }
    const float y = sMul[e] * d + sBia[e];
    if (g == 0) sY[ploc] = y;
    if (act) {
      gbf16x8 fB[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float* gc = gconst + 32 * c;
        fB[c] = gbf_basis8(y, gc, gc + 3 * GBF_K, gc + 2 * GBF_K);
        *reinterpret_cast<gbf16x8*>(GBF_SWZ(r0lo, r0hi, 32 * c) + 8 * g) = fB[c];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        gf32x4 hv[2];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          const int ft = 2 * u + hf;
          gf32x4 acc = {0.f, 0.f, 0.f, 0.f}, ua = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            const gbf16x8 wf = *reinterpret_cast<const gbf16x8*>(w2row + 16 * ft * GBF_W2S + 32 * c);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, oB[c], acc, 0, 0, 0);
            if constexpr (SPLIT_G) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, oL[c], acc, 0, 0, 0);
          }
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const gbf16x8 wf = *reinterpret_cast<const gbf16x8*>(w1row + 16 * ft * GBF_LW + 32 * c);
            ua = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, fB[c], ua, 0, 0, 0);
          }
          ua += *reinterpret_cast<const gf32x4*>(b1row + 16 * ft);
          {
            f32x2_t y0, g0, y1, g1;
            gelu_erf_both2(f32x2_t{ua[0], ua[1]}, y0, g0);
            gelu_erf_both2(f32x2_t{ua[2], ua[3]}, y1, g1);
            hv[hf] = gf32x4{y0[0], y0[1], y1[0], y1[1]};
            // gelu' passes through bf16 exactly where the unfused chain's saved tensor did
            const uint2 gb = gbf_pack4(gf32x4{g0[0], g0[1], g1[0], g1[1]});
            acc[0] *= __uint_as_float(gb.x << 16);
            acc[1] *= __uint_as_float(gb.x & 0xffff0000u);
            acc[2] *= __uint_as_float(gb.y << 16);
            acc[3] *= __uint_as_float(gb.y & 0xffff0000u);
          }
          *reinterpret_cast<uint2*>(GBF_SWZ(r1lo, r1hi, 16 * ft) + 4 * g) = gbf_pack4(acc);
          __builtin_amdgcn_sched_barrier(0);
        }
        float w[8] = {hv[0][0], hv[0][1], hv[0][2], hv[0][3], hv[1][0], hv[1][1], hv[1][2], hv[1][3]};
        hB[u] = gbf_pack8(w);      // slots 8g' + 4hf + e <-> features 32u + 16hf + 4g + e
      }
    } else {                       // tile past the end: its rows of the pair tiles contribute nothing
      const uint4 z4 = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        *reinterpret_cast<uint4*>(GBF_SWZ(r0lo, r0hi, 32 * c) + 8 * g) = z4;
        *reinterpret_cast<uint4*>(GBF_SWZ(r1lo, r1hi, 32 * c) + 8 * g) = z4;
        hB[c] = __builtin_bit_cast(gbf16x8, z4);
      }
    }
    __syncthreads();                                   // T0 = basis, T1 = du, sY = y of the workgroup's 128 pairs
    // ------------------------------------------------------------------------------------------------ phase B1
    // operand fragment (rows / columns fb*16 .., contraction = pairs 32s + 8g + 0..7) of a pair tile, transposing reads
#define GBF_TFRAG(P, S) gbf_tr8((P) + 32 * (S) * GBF_LW, (P) + (32 * (S) + 4) * GBF_LW)
#pragma unroll
    for (int sstep = 0; sstep < 4; ++sstep) {
      const gbf16x8 fa = GBF_TFRAG(t1w, sstep);                // du^T rows 16*wave ..
      aB1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, ones, aB1, 0, 0, 0);
#pragma unroll
      for (int kb = 0; kb < 8; ++kb) {
        const gbf16x8 fb = GBF_TFRAG(GBF_SWZ(t0lo, t0hi, 16 * kb), sstep);
        aW1[kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, aW1[kb], 0, 0, 0);
      }
    }
    {
      // dbasis^T[k = 16*wave + .., pair] = sum_f W1[f][k] du[pair][f]; A: W1^T fragment (contraction f = 32u + 8g + 0..7)
      gbf16x8 wfr[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        wfr[u] = gbf_tr8(w1tr + 32 * u * GBF_LW, w1tr + (32 * u + 4) * GBF_LW);
      }
      const gf32x4 kmu = *reinterpret_cast<const gf32x4*>(kconst);
      const gf32x4 kis = *reinterpret_cast<const gf32x4*>(kconst + GBF_K);
      const gf32x4 kcf = *reinterpret_cast<const gf32x4*>(kconst + 2 * GBF_K);
#pragma unroll 1                                           // (unrolled, the scheduler interleaves all 8 sub-tiles and spills 70 registers)
      for (int j = 0; j < 8; ++j) {
        gf32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const gbf16x8 dub = *reinterpret_cast<const gbf16x8*>(GBF_SWZ(d1lo, d1hi, 32 * u) + 16 * j * GBF_LW);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfr[u], dub, acc, 0, 0, 0);
        }
        const float yj = sY[16 * j + i];
        const gf32x2 yy = {yj, yj};
        gf32x2 dy2 = {0.f, 0.f};                          // (dbasis stays fp32 on its way into the Gaussian backward)
#define GBF_G2(LO, DV0, DV1)                                                                                   \
        {                                                                                                      \
          const gf32x2 is2 = {kis[LO], kis[LO + 1]};                                                           \
          const gf32x2 z = (yy - gf32x2{kmu[LO], kmu[LO + 1]}) * is2;                                          \
          const gf32x2 qq = (z * z) * (-0.72134752044448170f);           /* -log2(e) / 2 */                    \
          const gf32x2 val = gf32x2{__builtin_amdgcn_exp2f(qq[0]), __builtin_amdgcn_exp2f(qq[1])} * gf32x2{kcf[LO], kcf[LO + 1]}; \
          const gf32x2 t = gf32x2{DV0, DV1} * val;       /* (du, hence dbasis, is exactly 0 for pad pairs) */   \
          const gf32x2 zs = z * is2;                                                                           \
          const gf32x2 tz = t * zs;                                                                            \
          dy2 -= tz;                                                                                           \
          amu[LO] += tz[0]; amu[LO + 1] += tz[1];                                                              \
          const gf32x2 w = t * (z * zs - is2);                                                                 \
          asg[LO] += w[0]; asg[LO + 1] += w[1];                                                                \
        }
        GBF_G2(0, acc[0], acc[1])
        GBF_G2(2, acc[2], acc[3])
#undef GBF_G2
        float dy = dy2[0] + dy2[1];
        dy += __shfl_xor(dy, 16, 64);
        dy += __shfl_xor(dy, 32, 64);
        if (g == 0) atomicAdd(&sDy[16 * j + i], dy);
      }
    }
    __syncthreads();                                   // every wave has read basis / du; sDy complete
    // ------------------------------------------------------------------------------------------------ phase B2
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint4 hw = __builtin_bit_cast(uint4, hB[u]);
      *reinterpret_cast<uint2*>(GBF_SWZ(r0lo, r0hi, 32 * u) + 4 * g) = make_uint2(hw.x, hw.y);
      *reinterpret_cast<uint2*>(GBF_SWZ(r0lo, r0hi, 32 * u + 16) + 4 * g) = make_uint2(hw.z, hw.w);
    }
    *reinterpret_cast<gbf16x8*>(r1lo + 8 * g) = oB[0];              // dO: heads 32c + 8g ..
    *reinterpret_cast<gbf16x8*>(r1lo + 32 + 8 * g) = oB[1];
    if constexpr (SPLIT_G) {                                         // ... and its low parts in columns 64 + head
      *reinterpret_cast<gbf16x8*>(r1hi + 8 * g) = oL[0];
      *reinterpret_cast<gbf16x8*>(r1hi + 32 + 8 * g) = oL[1];
    }
    if (g == 0) {
      const float dyv = sDy[ploc];
      sDy[ploc] = 0.f;
      if (valid) {
        atomicAdd(&hist[e], dyv * d);
        atomicAdd(&hist[E + e], dyv);
      }
    }
    __syncthreads();
#pragma unroll
    for (int sstep = 0; sstep < 4; ++sstep) {
      const gbf16x8 fb = GBF_TFRAG(t0w, sstep);                // hidden columns 16*wave ..
#pragma unroll
      for (int hb = 0; hb < 4; ++hb) {
        const gbf16x8 fa = GBF_TFRAG(t1lo + 16 * hb, sstep);   // dO^T rows 16*hb .. (heads < 64: the low swizzle half)
        aW2[hb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, aW2[hb], 0, 0, 0);
        if constexpr (SPLIT_G) {
          const gbf16x8 fl = GBF_TFRAG(t1hi + 16 * hb, sstep);
          aW2[hb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fl, fb, aW2[hb], 0, 0, 0);
        }
      }
      // db2 of head block wave & 3 (waves 4..7 repeat it and drop it at the flush).  Unconditional on purpose: MFMA ignores
      // EXEC, a predicated one would still run -- on operands whose masked-off set-up did not.
      const gbf16x8 fo = GBF_TFRAG(t1lo + 16 * (wave & 3), sstep);
      aB2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fo, ones, aB2, 0, 0, 0);
      if constexpr (SPLIT_G) {
        const gbf16x8 fol = GBF_TFRAG(t1hi + 16 * (wave & 3), sstep);
        aB2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fol, ones, aB2, 0, 0, 0);
      }
    }
#undef GBF_TFRAG
#undef GBF_SWZ
    __syncthreads();                                   // tiles free for the next iteration
    act = n_act; valid = n_valid; d = n_d; e = n_e;
#pragma unroll
    for (int j = 0; j < 16; ++j) gv[j] = n_gv[j];
  }
  // ---- flush.  Accumulator tile: rows 4g + r, column i.  With a slab (the default: ops hands one in) every workgroup stores its
  // partial sums to its own [GBF_SLAB + 2 E] fp32 slot and gbf_slab_reduce_kernel folds the slots in a FIXED order -- bitwise
  // reproducible gradients; without one the partials meet in fp32 atomics (order-dependent in the last bits).
  float* const sl = slab ? slab + (long long)blockIdx.x * (GBF_SLAB + 2 * E) : nullptr;
#pragma unroll
  for (int kb = 0; kb < 8; ++kb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int o = (16 * wave + 4 * g + r) * GBF_K + 16 * kb + i;
      if (sl) sl[o] = aW1[kb][r]; else atomicAdd(dW1 + o, aW1[kb][r]);
    }
#pragma unroll
  for (int hb = 0; hb < 4; ++hb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int o = (16 * hb + 4 * g + r) * GBF_F + 16 * wave + i;
      if (sl) sl[GBF_F * GBF_K + o] = aW2[hb][r]; else atomicAdd(dW2 + o, aW2[hb][r]);
    }
  if (i == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int o = 16 * wave + 4 * g + r;
      if (sl) {
        sl[GBF_SLAB_B1 + o] = aB1[r];
        if (wave < 4) sl[GBF_SLAB_B2 + o] = aB2[r];
      } else {
        atomicAdd(db1 + o, aB1[r]);
        if (wave < 4) atomicAdd(db2 + o, aB2[r]);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float a = amu[r], sgv = asg[r];
    a += __shfl_xor(a, 1, 64); a += __shfl_xor(a, 2, 64); a += __shfl_xor(a, 4, 64); a += __shfl_xor(a, 8, 64);
    sgv += __shfl_xor(sgv, 1, 64); sgv += __shfl_xor(sgv, 2, 64); sgv += __shfl_xor(sgv, 4, 64); sgv += __shfl_xor(sgv, 8, 64);
    if (i == 0) {
      const int k = 16 * wave + 4 * g + r;
      const float ds = stds[k] < 0.f ? -sgv : sgv;        // d|std|/dstd
      if (sl) { sl[GBF_SLAB_MU + k] = a; sl[GBF_SLAB_SG + k] = ds; }
      else { atomicAdd(dmeans + k, a); atomicAdd(dstds + k, ds); }
    }
  }
  __syncthreads();
  for (int c = tid; c < E; c += 512) {
    const float a = hist[c], bb = hist[E + c];
    if (sl) { sl[GBF_SLAB + c] = a; sl[GBF_SLAB + E + c] = bb; }
    else {
      if (a != 0.f) atomicAdd(dmul + c, a);
      if (bb != 0.f) atomicAdd(dbias + c, bb);
    }
  }
}

// dst += sum over the workgroups' slab slots (no atomics, a FIXED order: four threads per gradient element each fold a contiguous
// quarter of the slots in slot order, then the quarters are added 0,1,2,3 -- the order depends on the grid size alone)
__global__ __launch_bounds__(256) void gbf_slab_reduce_kernel(const float* __restrict__ slab, int nwg, int E, float* __restrict__ dW1, float* __restrict__ db1,
                                                              float* __restrict__ dW2, float* __restrict__ db2, float* __restrict__ dmul,
                                                              float* __restrict__ dbias, float* __restrict__ dmeans, float* __restrict__ dstds) {
  __shared__ float part[4][64];
  const int SL = GBF_SLAB + 2 * E;
  const int el = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + el;
  const int per = (nwg + 3) / 4, w0 = q * per, w1 = min(nwg, w0 + per);
  float t = 0.f;
  if (e < SL) {
    const float* sp = slab + (long long)w0 * SL + e;
    int w = w0;
    for (; w + 8 <= w1; w += 8, sp += 8ll * SL) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = sp[(long long)j * SL];
#pragma unroll
      for (int j = 0; j < 8; ++j) t += v[j];
    }
    for (; w < w1; ++w, sp += SL) t += *sp;
  }
  part[q][el] = t;
  __syncthreads();
  if (q != 0 || e >= SL) return;
  t = ((part[0][el] + part[1][el]) + part[2][el]) + part[3][el];
  float* dst;
  if (e < GBF_F * GBF_K) dst = dW1 + e;
  else if (e < GBF_SLAB_B1) dst = dW2 + (e - GBF_F * GBF_K);
  else if (e < GBF_SLAB_B2) dst = db1 + (e - GBF_SLAB_B1);
  else if (e < GBF_SLAB_MU) dst = db2 + (e - GBF_SLAB_B2);
  else if (e < GBF_SLAB_SG) dst = dmeans + (e - GBF_SLAB_MU);
  else if (e < GBF_SLAB) dst = dstds + (e - GBF_SLAB_SG);
  else if (e < GBF_SLAB + E) dst = dmul + (e - GBF_SLAB);
  else dst = dbias + (e - GBF_SLAB - E);
  *dst += t;
}

// Tiled G (blocked rows, common.h) -> [B,N,N,H] bf16.  One block per (molecule, query block, key tile): a complete tile of a
// head is one contiguous KiB; the 16x16 pairs x H heads are regrouped in LDS so that each of the 16 query rows leaves as one
// contiguous run of 16 keys x H heads.
__global__ __launch_bounds__(256) void pair_untile_bwd_kernel(const float* __restrict__ g, bf16_t* __restrict__ out, int N, int H,
                                                              int nt) {
  extern __shared__ bf16_t ptile[];  // [256 pairs][H + 2]
  const int HP = H + 2;
  const int t = blockIdx.x % nt, qb = (blockIdx.x / nt) % nt, b = blockIdx.x / (nt * nt);
  const long long plane = pair_plane(N);
  const int vr = min(16, N - qb * 16);
  const float* src = g + (long long)b * H * plane + (long long)qb * 16 * pair_n4(N) + vr * 16 * t;
  for (int e = threadIdx.x; e < H * 64; e += 256) {      // one float4 (4 keys of one query) per step
    const int h = e >> 6, l = e & 63;
    const int q = l & 15, k0 = (l >> 4) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q < vr && t * 16 + k0 < N) v = *reinterpret_cast<const float4*>(src + (long long)h * plane + ((l >> 4) * vr + q) * 4);
    bf16_t* d = ptile + (q * 16 + k0) * HP + h;
    d[0] = f2bf(v.x); d[HP] = f2bf(v.y); d[2 * HP] = f2bf(v.z); d[3 * HP] = f2bf(v.w);
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 256 * (H / 2); e += 256) {   // two heads (4 bytes) per step
    const int pr = e / (H / 2), h2 = e - pr * (H / 2);
    const int i = qb * 16 + (pr >> 4), j = t * 16 + (pr & 15);
    if (i < N && j < N)
      *reinterpret_cast<uint32_t*>(out + (((long long)b * N + i) * N + j) * H + 2 * h2) =
          *reinterpret_cast<const uint32_t*>(ptile + pr * HP + 2 * h2);
  }
}

}  // namespace mmdti
using namespace mmdti;

static int gbf_check(const char* fn, long long P, int K, int E) {
  MMDTI_REQUIRE(P > 0 && K > 0 && K % 8 == 0 && E > 0, "%s: need P>0, K%%8==0, E>0 (K=%d)", fn, K);
  return MMDTI_OK;
}

extern "C" int mmdti_gbf_features_fwd(mmdti_stream_t stream, const float* dist, const long long* edge_type,
                                      const float* mul, const float* bias, const float* means, const float* stds,
                                      long long P, int K, int E, void* feat_bf16) {
  if (int e = gbf_check("gbf_features_fwd", P, K, E)) return e;
  MMDTI_REQUIRE(dist && edge_type && mul && bias && means && stds && feat_bf16, "gbf_features_fwd: null pointer");
  MMDTI_REQUIRE(aligned16(feat_bf16), "gbf_features_fwd: feat must be 16-byte aligned");
  long long blocks = (P + 15) / 16;
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(gbf_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dist, edge_type, mul,
                     bias, means, stds, P, K, E, (bf16_t*)feat_bf16);
  MMDTI_LAUNCH_CHECK();
  return MMDTI_OK;
}

extern "C" int mmdti_gbf_features_bwd(mmdti_stream_t stream, const float* dist, const long long* edge_type,
                                      const float* mul, const float* bias, const float* means, const float* stds,
                                      long long P, int K, int E, const void* dfeat_bf16, float* dmul, float* dbias,
                                      float* dmeans, float* dstds) {
  if (int e = gbf_check("gbf_features_bwd", P, K, E)) return e;
  MMDTI_REQUIRE(dist && edge_type && mul && bias && means && stds && dfeat_bf16 && dmul && dbias && dmeans && dstds,
                "gbf_features_bwd: null pointer");
  MMDTI_REQUIRE(aligned16(dfeat_bf16), "gbf_features_bwd: dfeat must be 16-byte aligned");
  long long blocks = (P + 15) / 16;
  if (blocks > 256 * 8) blocks = 256 * 8;
  const size_t smem = E <= GBF_MAXE ? 2 * (size_t)E * sizeof(float) : 0;
  hipLaunchKernelGGL(gbf_bwd_kernel, dim3((unsigned)blocks), dim3(256), smem, (hipStream_t)stream, dist, edge_type,
                     mul, bias, means, stds, P, K, E, (const bf16_t*)dfeat_bf16, dmul, dbias, dmeans, dstds);
  MMDTI_LAUNCH_CHECK();
  return MMDTI_OK;
}

extern "C" int mmdti_pair_permute_fwd(mmdti_stream_t stream, const float* x, float* out, int B, int N, int H, int ld) {
  MMDTI_REQUIRE(x && out && B > 0 && N > 0 && H > 0 && ld >= N, "pair_permute_fwd: bad arguments");
  const size_t smem = (size_t)N * (H + 1) * sizeof(float);
  MMDTI_REQUIRE(smem <= 64 * 1024, "pair_permute_fwd: N*H too large for the LDS tile (N=%d,H=%d)", N, H);
  hipLaunchKernelGGL(pair_permute_fwd_kernel, dim3(B * N), dim3(256), smem, (hipStream_t)stream, x, out, N, H, ld);
  MMDTI_LAUNCH_CHECK();
  return MMDTI_OK;
}

extern "C" int mmdti_pair_permute_bwd(mmdti_stream_t stream, const float* g, void* out_bf16, int B, int N, int H,
                                      int ld, int tiled) {
  MMDTI_REQUIRE(g && out_bf16 && B > 0 && N > 0 && H > 0 && ld >= N, "pair_permute_bwd: bad arguments");
  if (tiled && H % 2 == 0 && (size_t)256 * (H + 2) * 2 <= 64 * 1024 && aligned16(g)) {
    const int nt = (N + 15) / 16;
    hipLaunchKernelGGL(pair_untile_bwd_kernel, dim3(B * nt * nt), dim3(256), (size_t)256 * (H + 2) * 2, (hipStream_t)stream, g,
                       (bf16_t*)out_bf16, N, H, nt);
    MMDTI_LAUNCH_CHECK();
    return MMDTI_OK;
  }
  const size_t smem = (size_t)H * (N | 1) * sizeof(float);
  MMDTI_REQUIRE(smem <= 64 * 1024, "pair_permute_bwd: N*H too large for the LDS tile (N=%d,H=%d)", N, H);
  hipLaunchKernelGGL(pair_permute_bwd_kernel, dim3(B * N), dim3(256), smem, (hipStream_t)stream, g, (bf16_t*)out_bf16,
                     N, H, ld, tiled);
  MMDTI_LAUNCH_CHECK();
  return MMDTI_OK;
}

static int edge_bytes_ok(int eb) { return eb == 8 || eb == 4 || eb == 2; }

extern "C" int mmdti_gbf_bias_fwd(mmdti_stream_t stream, const float* dist, const void* edge_type, int edge_bytes, const float* mul,
                                  const float* bias, const float* means, const float* stds, const void* w1_bf16,
                                  const float* b1, const void* w2_bf16, const float* b2, int B, int N, int ld, int K, int F,
                                  int H, int E, void* out, void* feat_bf16, void* u_bf16, void* h_bf16, int flags,
                                  const int* tile_prefix, const int* row_blocks) {
  // bit 0: tiled pair layout; bit 1: u_bf16 receives gelu'(u) instead of u; bit 2: compact planes (out is fp16; tiled only)
  const int tiled = flags & 1, ugrad = (flags >> 1) & 1, compact = (flags >> 2) & 1;
  MMDTI_REQUIRE(!compact || tiled, "gbf_bias_fwd: compact planes (flags bit 2) exist in the tiled layout only");
  MMDTI_REQUIRE(!tile_prefix || tiled, "gbf_bias_fwd: tile_prefix (ragged batches) needs the tiled pair layout");
  MMDTI_REQUIRE(!row_blocks || tile_prefix, "gbf_bias_fwd: row_blocks (packed token rows) comes with tile_prefix");
  MMDTI_REQUIRE(dist && edge_type && mul && bias && means && stds && w1_bf16 && b1 && w2_bf16 && b2 && out, "gbf_bias_fwd: null argument");
  MMDTI_REQUIRE(edge_bytes_ok(edge_bytes), "gbf_bias_fwd: edge types must be int64, int32 or int16 (edge_bytes=%d)", edge_bytes);
  MMDTI_REQUIRE(K == GBF_K && F == GBF_F && H == GBF_H, "gbf_bias_fwd: built for %d gaussians, %d hidden, %d heads (got %d,%d,%d)",
                GBF_K, GBF_F, GBF_H, K, F, H);
  MMDTI_REQUIRE(B > 0 && N > 0 && ld >= N && ld % 4 == 0 && E > 0, "gbf_bias_fwd: bad shape");
  MMDTI_REQUIRE(aligned16(w1_bf16) && aligned16(w2_bf16) && aligned16(out), "gbf_bias_fwd: 16-byte alignment required");
  const bool save = feat_bf16 || u_bf16 || h_bf16;
  MMDTI_REQUIRE(!save || (feat_bf16 && u_bf16 && h_bf16 && aligned16(feat_bf16) && aligned16(u_bf16) && aligned16(h_bf16)),
                "gbf_bias_fwd: the three saved intermediates come together, 16-byte aligned");
  const int nb4 = (N + 3) / 4;      // tiled planes: the 4x4 blocks that hold a real pair
  const int tpm = tiled ? nb4 * nb4 : cdiv((long long)N * ld, 16);
  const long long ntiles = (long long)B * tpm;
  MMDTI_REQUIRE(ntiles < (1ll << 30) && (long long)N * N < (1ll << 30), "gbf_bias_fwd: batch too large for 32-bit tile indices");
  const int grid = (int)((ntiles + 7) / 8 < 512 ? (ntiles + 7) / 8 : 512);     // two resident workgroups per CU
  const int ltab = E <= GBF_FWD_MAXE ? 1 : 0;
  const size_t smem = (size_t)(GBF_F + GBF_H) * GBF_WS * 2 + (size_t)(3 * GBF_K + GBF_F + GBF_H + (ltab ? 2 * E : 0)) * 4;
  static bool attr_done = false;
  if (!attr_done) {
    const int cap = (int)((size_t)(GBF_F + GBF_H) * GBF_WS * 2 + (size_t)(3 * GBF_K + GBF_F + GBF_H + 2 * GBF_FWD_MAXE) * 4);
    const void* fns[6] = {reinterpret_cast<const void*>(gbf_bias_fwd_kernel<true, true, float>), reinterpret_cast<const void*>(gbf_bias_fwd_kernel<true, false, float>),
                          reinterpret_cast<const void*>(gbf_bias_fwd_kernel<false, true, float>), reinterpret_cast<const void*>(gbf_bias_fwd_kernel<false, false, float>),
                          reinterpret_cast<const void*>(gbf_bias_fwd_kernel<true, true, _Float16>), reinterpret_cast<const void*>(gbf_bias_fwd_kernel<false, true, _Float16>)};
    for (int f = 0; f < 6; ++f)
      if (hipFuncSetAttribute(fns[f], hipFuncAttributeMaxDynamicSharedMemorySize, cap) != hipSuccess) {
        set_error("gbf_bias_fwd: hipFuncSetAttribute failed");
        return MMDTI_ERR_LAUNCH;
      }
    attr_done = true;
  }
#define GBF_L(SAVE, TILED, OT)                                                                                                      \
  hipLaunchKernelGGL((gbf_bias_fwd_kernel<SAVE, TILED, OT>), dim3(grid), dim3(512), smem, (hipStream_t)stream, dist, edge_type, edge_bytes, mul, bias, \
                     means, stds, (const bf16_t*)w1_bf16, b1, (const bf16_t*)w2_bf16, b2, (OT*)out, (bf16_t*)feat_bf16, (bf16_t*)u_bf16, \
                     (bf16_t*)h_bf16, B, N, ld, E, tpm, ugrad, ltab, tile_prefix, row_blocks)
  if (compact) { if (save) GBF_L(true, true, _Float16); else GBF_L(false, true, _Float16); }
  else if (save) { if (tiled) GBF_L(true, true, float); else GBF_L(true, false, float); }
  else           { if (tiled) GBF_L(false, true, float); else GBF_L(false, false, float); }
#undef GBF_L
  MMDTI_LAUNCH_CHECK();
  return MMDTI_OK;
}

extern "C" int mmdti_gbf_bias_bwd(mmdti_stream_t stream, const void* g, const float* dist, const void* edge_type, int edge_bytes,
                                  const float* mul, const float* bias, const float* means, const float* stds,
                                  const void* w1_bf16, const void* w2_bf16, const void* u_bf16, int B, int N, int ld, int K,
                                  int F, int H, int E, int flags, void* do_bf16, void* du_bf16, float* dmul, float* dbias,
                                  float* dmeans, float* dstds) {
  // bit 0: tiled pair layout; bit 1: u_bf16 holds gelu'(u); bit 2: compact planes (g is bf16; tiled only)
  const int tiled = flags & 1, ugrad = (flags >> 1) & 1, compact = (flags >> 2) & 1;
  MMDTI_REQUIRE(!compact || tiled, "gbf_bias_bwd: compact planes (flags bit 2) exist in the tiled layout only");
  MMDTI_REQUIRE(g && dist && edge_type && mul && bias && means && stds && w1_bf16 && w2_bf16 && u_bf16 && do_bf16 && du_bf16 && dmul &&
                    dbias && dmeans && dstds, "gbf_bias_bwd: null argument");
  MMDTI_REQUIRE(K == GBF_K && F == GBF_F && H == GBF_H, "gbf_bias_bwd: built for %d gaussians, %d hidden, %d heads (got %d,%d,%d)",
                GBF_K, GBF_F, GBF_H, K, F, H);
  MMDTI_REQUIRE(B > 0 && N > 0 && ld >= N && ld % 4 == 0 && E > 0 && E <= GBF_MAXE, "gbf_bias_bwd: bad shape (E <= %d)", GBF_MAXE);
  MMDTI_REQUIRE(edge_bytes_ok(edge_bytes), "gbf_bias_bwd: edge types must be int64, int32 or int16 (edge_bytes=%d)", edge_bytes);
  MMDTI_REQUIRE(aligned16(u_bf16) && aligned16(do_bf16) && aligned16(du_bf16), "gbf_bias_bwd: 16-byte alignment required");
  const int nb4 = (N + 3) / 4;      // tiled planes: the 4x4 blocks that hold a real pair
  const int tpm = tiled ? nb4 * nb4 : cdiv((long long)N * ld, 16);
  const long long ntiles = (long long)B * tpm;
  const int grid = (int)(ntiles / 4 + 1 < 1024 ? ntiles / 4 + 1 : 1024);
  const size_t smem = (size_t)GBF_K * GBF_WS * 2 + (size_t)GBF_F * GBF_W2S * 2 + (size_t)(3 * GBF_K + 4 * E) * 4;
  MMDTI_REQUIRE(smem <= 96 * 1024, "gbf_bias_bwd: %d edge types do not fit the LDS tables", E);
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(gbf_bias_bwd_kernel<true, float>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(gbf_bias_bwd_kernel<false, float>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(gbf_bias_bwd_kernel<true, __bf16>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) != hipSuccess) {
      set_error("gbf_bias_bwd: hipFuncSetAttribute failed");
      return MMDTI_ERR_LAUNCH;
    }
    attr_done = true;
  }
#define GBF_B(TILED, GT)                                                                                                      \
  hipLaunchKernelGGL((gbf_bias_bwd_kernel<TILED, GT>), dim3(grid), dim3(256), smem, (hipStream_t)stream, (const GT*)g, dist, edge_type, edge_bytes, mul, bias, \
                     means, stds, (const bf16_t*)w1_bf16, (const bf16_t*)w2_bf16, (const bf16_t*)u_bf16, (bf16_t*)do_bf16,    \
                     (bf16_t*)du_bf16, dmul, dbias, dmeans, dstds, B, N, ld, E, tpm, ugrad)
  if (compact) GBF_B(true, __bf16); else if (tiled) GBF_B(true, float); else GBF_B(false, float);
#undef GBF_B
  MMDTI_LAUNCH_CHECK();
  return MMDTI_OK;
}

extern "C" int mmdti_gbf_bias_bwd_full(mmdti_stream_t stream, const void* g, const float* dist, const void* edge_type, int edge_bytes,
                                       const float* mul, const float* bias, const float* means, const float* stds, const void* w1_bf16,
                                       const float* b1, const void* w2_bf16, int B, int N, int ld, int K, int F, int H, int E, int flags,
                                       float* dw1, float* db1, float* dw2, float* db2, float* dmul, float* dbias, float* dmeans,
                                       float* dstds, const int* tile_prefix, const int* row_blocks, void* workspace, long long workspace_bytes) {
  const int tiled = flags & 1, compact = (flags >> 2) & 1;   // bit 0: tiled pair layout; bit 2: compact planes (g is bf16; tiled only)
  MMDTI_REQUIRE(!compact || tiled, "gbf_bias_bwd_full: compact planes (flags bit 2) exist in the tiled layout only");
  MMDTI_REQUIRE(!tile_prefix || tiled, "gbf_bias_bwd_full: tile_prefix (ragged batches) needs the tiled pair layout");
  MMDTI_REQUIRE(!row_blocks || tile_prefix, "gbf_bias_bwd_full: row_blocks (packed token rows) comes with tile_prefix");
  MMDTI_REQUIRE(g && dist && edge_type && mul && bias && means && stds && w1_bf16 && b1 && w2_bf16 && dw1 && db1 && dw2 && db2 && dmul && dbias &&
                    dmeans && dstds, "gbf_bias_bwd_full: null argument");
  MMDTI_REQUIRE(edge_bytes_ok(edge_bytes), "gbf_bias_bwd_full: edge types must be int64, int32 or int16 (edge_bytes=%d)", edge_bytes);
  MMDTI_REQUIRE(K == GBF_K && F == GBF_F && H == GBF_H, "gbf_bias_bwd_full: built for %d gaussians, %d hidden, %d heads (got %d,%d,%d)",
                GBF_K, GBF_F, GBF_H, K, F, H);
  MMDTI_REQUIRE(B > 0 && N > 0 && ld >= N && ld % 4 == 0 && E > 0 && E <= GBF_FULL_MAXE, "gbf_bias_bwd_full: bad shape (E <= %d)", GBF_FULL_MAXE);
  MMDTI_REQUIRE((long long)N * ld * GBF_H < (1ll << 29) && (long long)B * N * ld / 16 < (1ll << 30), "gbf_bias_bwd_full: batch too large for 32-bit tile offsets");
  MMDTI_REQUIRE(aligned16(w1_bf16), "gbf_bias_bwd_full: W1 must be 16-byte aligned");
  const int nb = (N + 3) / 4;
  const int tpm = tiled ? nb * nb : cdiv((long long)N * ld, 16);
  const long long ntiles = (long long)B * tpm;
  const int grid = (int)((ntiles + 7) / 8 < 256 ? (ntiles + 7) / 8 : 256);
  const size_t smem = gbf_full_smem(E);
  static bool attr_done = false;
  if (!attr_done) {
    const int cap = (int)gbf_full_smem(GBF_FULL_MAXE);
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(gbf_bias_bwd_full_kernel<true, float>), hipFuncAttributeMaxDynamicSharedMemorySize, cap) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(gbf_bias_bwd_full_kernel<false, float>), hipFuncAttributeMaxDynamicSharedMemorySize, cap) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(gbf_bias_bwd_full_kernel<true, __bf16>), hipFuncAttributeMaxDynamicSharedMemorySize, cap) != hipSuccess) {
      set_error("gbf_bias_bwd_full: hipFuncSetAttribute failed");
      return MMDTI_ERR_LAUNCH;
    }
    attr_done = true;
  }
#define GBF_FB(TILED, GT)                                                                                                          \
  hipLaunchKernelGGL((gbf_bias_bwd_full_kernel<TILED, GT>), dim3(grid), dim3(512), smem, (hipStream_t)stream, (const GT*)g, dist, edge_type, edge_bytes, mul, \
                     bias, means, stds, (const bf16_t*)w1_bf16, b1, (const bf16_t*)w2_bf16, dw1, db1, dw2, db2, dmul, dbias, dmeans, dstds, B, \
                     N, ld, E, tpm, tile_prefix, row_blocks, slab)
  // (a workspace of at least grid slots: partial sums in per-workgroup slabs + a fixed-order reduce; else fp32 atomics)
  const long long SL = GBF_SLAB + 2 * (long long)E;
  float* slab = (workspace && aligned16(workspace) && workspace_bytes >= (long long)grid * SL * 4) ? reinterpret_cast<float*>(workspace) : nullptr;
  if (compact) GBF_FB(true, __bf16); else if (tiled) GBF_FB(true, float); else GBF_FB(false, float);
#undef GBF_FB
  if (slab)
    hipLaunchKernelGGL(gbf_slab_reduce_kernel, dim3(cdiv(SL, 64)), dim3(256), 0, (hipStream_t)stream, slab, grid, E, dw1, db1, dw2, db2, dmul, dbias, dmeans, dstds);
  MMDTI_LAUNCH_CHECK();
  return MMDTI_OK;
}

/* bytes of workspace that make mmdti_gbf_bias_bwd_full reproducible (one slab per workgroup of its persistent grid, at most 256) */
extern "C" int mmdti_gbf_bias_bwd_full_workspace(int E) { return (int)(256ll * (GBF_SLAB + 2ll * (E > 0 ? E : 0)) * 4); }
