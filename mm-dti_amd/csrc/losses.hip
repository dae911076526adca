// Batch-coupled heads of the fine-tune step: InfoNCE (models/infonce.py), ConR / SupCon (models/contrastive.py),
// FDS (models/fds.py, utils/util.py:159-169), the classification head (models/mm_model.py:44-84) and the task
// losses (models/nnmodel.py:24-34).  All are B x B or B x D problems (B = molecules per rank, <= a few thousand):
// latency-bound, so each is a small number of single-pass kernels with the gradient produced in the same pass as
// the loss (no materialised autograd graph, no host sync, no Python loops over B as in the reference).
#include "common.h"

namespace mmdti {

__device__ __forceinline__ float block_sum(float v, float* red) {  // 256 threads
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// ---------------------------------------------------------------- unmasked sequence mean (infonce.py:32-33)
__global__ __launch_bounds__(256) void seq_mean_fwd_kernel(const bf16_t* __restrict__ x, int S, int D, int ld, float* __restrict__ out) {
  const int b = blockIdx.x;
  for (int d = threadIdx.x; d < D; d += 256) {
    float s = 0.f;
    for (int i = 0; i < S; ++i) s += bf2f(x[((long long)b * S + i) * ld + d]);
    out[(long long)b * D + d] = s / (float)S;
  }
}
__global__ __launch_bounds__(256) void seq_mean_bwd_kernel(const float* __restrict__ dout, int S, int D, int ld, bf16_t* __restrict__ dx) {
  const long long row = blockIdx.x;  // b*S + s
  const int b = (int)(row / S);
  for (int d = threadIdx.x; d < ld; d += 256)
    dx[row * ld + d] = f2bf(d < D ? dout[(long long)b * D + d] / (float)S : 0.f);
}

// Wide rows (D % 8 == 0): 16-byte loads -- the InfoNCE head pools the 512-wide GELU outputs of every token (pool first, project
// after: mean_t(W2 h_t + b2) = W2 mean_t(h_t) + b2).  One workgroup per (sequence, 64-column slice): 32 row phases x 8 pieces, the
// phases folded through LDS in a FIXED order.  (Rounds 1-3 split a sequence's rows over workgroups that met in fp32 atomics: the
// pooled embedding -- a forward activation -- differed by an ulp from run to run, and every gradient of the step with it; found as
// the source of the run-to-run noise in the step's ill-conditioned gradient sums, DESIGN.md "determinism".)
// row_off / n_real non-null: packed token rows (see mmdti_seq_mean_packed_fwd) -- the representative pad row weighted by the number
// of padded positions it stands for.
__global__ __launch_bounds__(256) void seq_mean_fwd_vec_kernel(const bf16_t* __restrict__ x, int S, int D, int ld, const int* __restrict__ row_off,
                                                               const int* __restrict__ n_real, float* __restrict__ out) {
  __shared__ float red[32][65];
  const int b = blockIdx.x, c0 = blockIdx.y * 64;
  const int piece = threadIdx.x & 7, phase = threadIdx.x >> 3;
  long long r0 = (long long)b * S;
  int rows = S, nr = S;
  float wpad = 1.f;
  if (row_off) {
    r0 = row_off[b];
    rows = row_off[b + 1] - row_off[b];
    nr = n_real[b];
    wpad = (float)(S - nr);
  }
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const int col = c0 + piece * 8;
  if (col < D) {
    for (int r = phase; r < rows; r += 32) {
      const uint4 u = *reinterpret_cast<const uint4*>(x + (r0 + r) * ld + col);
      const uint32_t w4[4] = {u.x, u.y, u.z, u.w};
      const float w = r < nr ? 1.f : wpad;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc[2 * e] += w * __uint_as_float(w4[e] << 16);
        acc[2 * e + 1] += w * __uint_as_float(w4[e] & 0xffff0000u);
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[phase][piece * 8 + e] = acc[e];
  __syncthreads();
  if (threadIdx.x < 64 && c0 + threadIdx.x < D) {
    float t = 0.f;
#pragma unroll
    for (int ph = 0; ph < 32; ++ph) t += red[ph][threadIdx.x];
    out[(long long)b * D + c0 + threadIdx.x] = t * (1.0f / (float)S);
  }
}
// dx[row, d] = bf16(dout[b, d] / S * f(aux[row, d])) with f = 1 (aux_mode 0), aux itself (1: a saved gelu') or gelu'(aux) (2)
__global__ __launch_bounds__(256) void seq_mean_bwd_vec_kernel(const float* __restrict__ dout, int S, int D, int ld, const bf16_t* __restrict__ aux,
                                                               int ld_aux, int aux_mode, bf16_t* __restrict__ dx, long long rows) {
  const int cpr = ld >> 3;
  const long long chunk = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long row = chunk / cpr;
  if (row >= rows) return;
  const int c8 = (int)(chunk - row * cpr) * 8;
  const int b = (int)(row / S);
  const float inv = 1.0f / (float)S;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = (c8 + e < D) ? dout[(long long)b * D + c8 + e] * inv : 0.f;
  if (aux_mode) {
    const uint4 u = *reinterpret_cast<const uint4*>(aux + row * ld_aux + c8);
    const uint32_t w4[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float a = __uint_as_float((e & 1) ? (w4[e >> 1] & 0xffff0000u) : (w4[e >> 1] << 16));
      v[e] *= aux_mode == 1 ? a : gelu_erf_grad(a);
    }
  }
  uint4 o;
  o.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
  o.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
  o.z = (uint32_t)f2bf(v[4]) | ((uint32_t)f2bf(v[5]) << 16);
  o.w = (uint32_t)f2bf(v[6]) | ((uint32_t)f2bf(v[7]) << 16);
  *reinterpret_cast<uint4*>(dx + row * ld + c8) = o;
}

__global__ __launch_bounds__(256) void seq_mean_packed_bwd_kernel(const float* __restrict__ dout, int S, int D, int ld, const int* __restrict__ row_off,
                                                                  const int* __restrict__ n_real, const int* __restrict__ row_seq,
                                                                  const bf16_t* __restrict__ aux, int ld_aux, int aux_mode, bf16_t* __restrict__ dx,
                                                                  long long rows) {
  const int cpr = ld >> 3;
  const long long chunk = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long row = chunk / cpr;
  if (row >= rows) return;
  const int c8 = (int)(chunk - row * cpr) * 8;
  const int b = row_seq[row];
  const int nr = n_real[b];
  const float w = ((int)row - row_off[b] < nr ? 1.f : (float)(S - nr)) / (float)S;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = (c8 + e < D) ? dout[(long long)b * D + c8 + e] * w : 0.f;
  if (aux_mode) {
    const uint4 u = *reinterpret_cast<const uint4*>(aux + row * ld_aux + c8);
    const uint32_t w4[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float a = __uint_as_float((e & 1) ? (w4[e >> 1] & 0xffff0000u) : (w4[e >> 1] << 16));
      v[e] *= aux_mode == 1 ? a : gelu_erf_grad(a);
    }
  }
  uint4 o;
  o.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
  o.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
  o.z = (uint32_t)f2bf(v[4]) | ((uint32_t)f2bf(v[5]) << 16);
  o.w = (uint32_t)f2bf(v[6]) | ((uint32_t)f2bf(v[7]) << 16);
  *reinterpret_cast<uint4*>(dx + row * ld + c8) = o;
}

// ---------------------------------------------------------------- F.normalize (eps 1e-12)
__global__ __launch_bounds__(64) void l2norm_fwd_kernel(const float* __restrict__ x, int D, int ldx, float* __restrict__ xh, float* __restrict__ invn) {
  const int b = blockIdx.x, lane = threadIdx.x;
  float s = 0.f;
  for (int d = lane; d < D; d += 64) { float v = x[(long long)b * ldx + d]; s += v * v; }
  const float inv = 1.0f / fmaxf(sqrtf(wave_sum(s)), 1e-12f);
  for (int d = lane; d < D; d += 64) xh[(long long)b * D + d] = x[(long long)b * ldx + d] * inv;
  if (lane == 0) invn[b] = inv;
}
// dx = inv * (dxh - xh * <dxh, xh>)
__global__ __launch_bounds__(64) void l2norm_bwd_kernel(const float* __restrict__ dxh, const float* __restrict__ xh, const float* __restrict__ invn,
                                                        int D, int ldx, float* __restrict__ dx) {
  const int b = blockIdx.x, lane = threadIdx.x;
  float s = 0.f;
  for (int d = lane; d < D; d += 64) s += dxh[(long long)b * D + d] * xh[(long long)b * D + d];
  s = wave_sum(s);
  const float inv = invn[b];
  for (int d = lane; d < D; d += 64) dx[(long long)b * ldx + d] = inv * (dxh[(long long)b * D + d] - xh[(long long)b * D + d] * s);
}

// ---------------------------------------------------------------- InfoNCE, one direction (infonce.py:93-98)
// Two small kernels, no atomics on the gradients:
//  rows: block per anchor i (global index row0+i): logits_j = <q_i, k_j>/T over all Bg keys, CE with label i; writes the
//        logit-gradient row gl[i][:] to scratch and dq_i += sum_j gl_ij k_j (thread = feature column).
//  cols: block per key j: dk_j += sum_i gl_ij q_i.
__global__ __launch_bounds__(256) void infonce_rows_kernel(const float* __restrict__ q, const float* __restrict__ k, int Bg, int D, int row0,
                                                           float invT, float* __restrict__ loss_sum, float* __restrict__ dq,
                                                           float* __restrict__ gl_out) {
  extern __shared__ float sm[];  // qrow[D] | logits[Bg] | red[4]
  float* qrow = sm;
  float* lg = sm + D;
  float* red = lg + Bg;
  const int il = blockIdx.x, i = row0 + il;
  for (int d = threadIdx.x; d < D; d += 256) qrow[d] = q[(long long)i * D + d];
  __syncthreads();
  float m = -INFINITY;
  for (int j = threadIdx.x; j < Bg; j += 256) {
    float s = 0.f;
    for (int d = 0; d < D; ++d) s += qrow[d] * k[(long long)j * D + d];
    s *= invT;
    lg[j] = s;
    m = fmaxf(m, s);
  }
  m = block_max(m, red);
  float se = 0.f;
  for (int j = threadIdx.x; j < Bg; j += 256) se += __expf(lg[j] - m);
  se = block_sum(se, red);
  const float lse = m + __logf(se);
  if (threadIdx.x == 0) gl_out[(long long)gridDim.x * Bg + il] = lse - lg[i];     // (this row's loss term: summed in row order by the column pass)
  const float gscale = 0.5f / (float)Bg;  // d[(CE_a + CE_b)/2 mean over Bg]/d CE_i
  __syncthreads();
  for (int j = threadIdx.x; j < Bg; j += 256) {
    const float gl = (__expf(lg[j] - lse) - (j == i ? 1.f : 0.f)) * gscale * invT;
    lg[j] = gl;
    gl_out[(long long)il * Bg + j] = gl;
  }
  __syncthreads();
  for (int d = threadIdx.x; d < D; d += 256) {
    float acc = 0.f;
    for (int j = 0; j < Bg; ++j) acc += lg[j] * k[(long long)j * D + d];
    dq[(long long)i * D + d] += acc;
  }
}

__global__ __launch_bounds__(64) void infonce_cols_kernel(const float* __restrict__ q, const float* __restrict__ gl, int Bg, int D, int row0,
                                                          int Bl, float* __restrict__ dk, float* __restrict__ loss_sum) {
  const int j = blockIdx.x;
  if (j == 0 && threadIdx.x == 0) {      // loss_sum += the rows' loss terms, in row order (one writer: reproducible to the bit)
    float t = 0.f;
    for (int il = 0; il < Bl; ++il) t += gl[(long long)Bl * Bg + il];
    *loss_sum += t;
  }
  for (int d = threadIdx.x; d < D; d += 64) {
    float acc = 0.f;
    for (int il = 0; il < Bl; ++il) acc += gl[(long long)il * Bg + j] * q[(long long)(row0 + il) * D + d];
    dk[(long long)j * D + d] += acc;
  }
}

// ---------------------------------------------------------------- InfoNCE on the matrix pipe (feature width <= 64)
// The B x B_global similarity matrix and both gradient products as fp32 MFMAs (v_mfma_f32_16x16x4_f32: fp32 products, fp32
// accumulation -- the loss keeps fp32 precision; bf16 operands would cost 1e-3 on a temperature-0.1 softmax):
//   rows kernel, one workgroup per 16 anchors, the waves share the key tiles:
//     pass 1  L^tile = Q_a . K_t^T  (13 k-steps at D = 50)  ->  online (max, sum) per anchor
//     pass 2  L^tile again -> gl = (softmax - onehot) * scale -> scratch;  dQ_a += gl . K_t  (gl transposed through a 16 x 17
//             LDS patch per wave: the accumulator layout has anchors on registers, the A operand wants them on lanes)
//   cols kernel, one workgroup per 16 keys:  dK_t += gl^T . Q  over the rank's anchors.
// At the 8-GPU global batch (2048 keys) the scalar kernels below walk 2048-long serial loops per thread; these do not.
typedef float nce_f32x4 __attribute__((ext_vector_type(4)));
#define NCE_MFMA(A, B, C) __builtin_amdgcn_mfma_f32_16x16x4f32(A, B, C, 0, 0, 0)
constexpr int NCE_MAXD = 64;

__device__ __forceinline__ void nce_combine(float& m, float& s, float m2, float s2) {
  const float mn = fmaxf(m, m2);
  if (mn == -INFINITY) { m = mn; s = 0.f; return; }
  s = s * __expf(m - mn) + s2 * __expf(m2 - mn);
  m = mn;
}

__global__ __launch_bounds__(256) void infonce_rows_mfma_kernel(const float* __restrict__ q, const float* __restrict__ k, int Bg, int D, int row0,
                                                                int Bl, float invT, float* __restrict__ loss_sum, float* __restrict__ dq,
                                                                float* __restrict__ gl_out) {
  extern __shared__ float sm[];
  const int DP = (D + 3) & ~3;
  float* sQ = sm;                        // [16][DP]
  float* sStat = sQ + 16 * DP;           // [4 waves][16 anchors][2]
  float* sLse = sStat + 4 * 16 * 2;      // [16]
  float* patch = sLse + 16;              // [4 waves][16][17]
  float* sDq = patch + 4 * 16 * 17;      // [4 waves][16 anchors][64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, i = lane & 15;
  const int il0 = blockIdx.x * 16;
  for (int c = tid; c < 16 * DP; c += 256) {
    const int a = c / DP, d = c - a * DP;
    sQ[c] = (il0 + a < Bl && d < D) ? q[(long long)(row0 + il0 + a) * D + d] : 0.f;
  }
  __syncthreads();
  const int nkt = (Bg + 15) >> 4, nks = DP >> 2, NU = (D + 15) >> 4;
  auto logits = [&](int t) -> nce_f32x4 {      // rows = anchors 4g + r, column = key 16t + i (unscaled)
    const int key = 16 * t + i;
    const float* kr = k + (long long)(key < Bg ? key : 0) * D;
    nce_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int s4 = 0; s4 < nks; ++s4) {
      const int d = 4 * s4 + g;
      const float bv = (key < Bg && d < D) ? kr[d] : 0.f;
      acc = NCE_MFMA(sQ[i * DP + d], bv, acc);
    }
    return acc;
  };
  // ---- pass 1: row max / sum
  float m[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY}, sum[4] = {0.f, 0.f, 0.f, 0.f};
  for (int t = wave; t < nkt; t += 4) {
    const nce_f32x4 acc = logits(t);
    if (16 * t + i < Bg) {
#pragma unroll
      for (int r = 0; r < 4; ++r) nce_combine(m[r], sum[r], acc[r] * invT, 1.0f);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
      const float m2 = __shfl_xor(m[r], o, 64), s2 = __shfl_xor(sum[r], o, 64);
      nce_combine(m[r], sum[r], m2, s2);
    }
    if (i == 0) {
      sStat[(wave * 16 + 4 * g + r) * 2] = m[r];
      sStat[(wave * 16 + 4 * g + r) * 2 + 1] = sum[r];
    }
  }
  __syncthreads();
  if (tid < 16) {
    float mm = -INFINITY, ss = 0.f;
    for (int w = 0; w < 4; ++w) nce_combine(mm, ss, sStat[(w * 16 + tid) * 2], sStat[(w * 16 + tid) * 2 + 1]);
    sLse[tid] = mm + __logf(ss);
  }
  __syncthreads();
  // ---- pass 2: logit gradients, loss, dQ
  const float gscale = 0.5f / (float)Bg;     // d[(CE_a + CE_b)/2 mean over Bg]/d CE_i
  nce_f32x4 dacc[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) dacc[u] = nce_f32x4{0.f, 0.f, 0.f, 0.f};
  float* pw = patch + wave * 16 * 17;
  for (int t = wave; t < nkt; t += 4) {
    const nce_f32x4 acc = logits(t);
    const int key = 16 * t + i;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int a = 4 * g + r, il = il0 + a;
      float gl = 0.f;
      if (key < Bg && il < Bl) {
        const float l = acc[r] * invT, lse = sLse[a];
        const bool diag = key == row0 + il;
        gl = (__expf(l - lse) - (diag ? 1.f : 0.f)) * gscale * invT;
        gl_out[(long long)il * Bg + key] = gl;
        if (diag) gl_out[(long long)Bl * Bg + il] = lse - l;      // (this row's loss term: summed in row order by the column pass)
      }
      pw[a * 17 + i] = gl;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int sp = 0; sp < 4; ++sp) {
      const float av = pw[i * 17 + 4 * sp + g];                 // A: anchor i, key 4 sp + g of the tile
      const int kk = 16 * t + 4 * sp + g;
      const float* kr = k + (long long)(kk < Bg ? kk : 0) * D;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (u < NU) {
          const int d = 16 * u + i;
          const float bv = (kk < Bg && d < D) ? kr[d] : 0.f;     // B: key 4 sp + g, feature 16 u + i
          dacc[u] = NCE_MFMA(av, bv, dacc[u]);
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();                              // the next tile overwrites the patch
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int r = 0; r < 4; ++r) sDq[(wave * 16 + 4 * g + r) * 64 + 16 * u + i] = dacc[u][r];
  __syncthreads();
  for (int c = tid; c < 16 * 64; c += 256) {
    const int a = c >> 6, d = c & 63;
    if (il0 + a < Bl && d < D)
      dq[(long long)(row0 + il0 + a) * D + d] += sDq[c] + sDq[16 * 64 + c] + sDq[2 * 16 * 64 + c] + sDq[3 * 16 * 64 + c];
  }
}

__global__ __launch_bounds__(256) void infonce_cols_mfma_kernel(const float* __restrict__ q, const float* __restrict__ gl, int Bg, int D, int row0,
                                                                int Bl, float* __restrict__ dk, float* __restrict__ loss_sum) {
  __shared__ float sDk[4 * 16 * 64];
  if (blockIdx.x == 0 && threadIdx.x == 0) {      // loss_sum += the rows' loss terms, in row order (one writer: reproducible to the bit)
    float t = 0.f;
    for (int il = 0; il < Bl; ++il) t += gl[(long long)Bl * Bg + il];
    *loss_sum += t;
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, i = lane & 15;
  const int j0 = blockIdx.x * 16, NU = (D + 15) >> 4;
  const int key = j0 + i;
  nce_f32x4 dacc[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) dacc[u] = nce_f32x4{0.f, 0.f, 0.f, 0.f};
  const int nat = (Bl + 15) >> 4;
  for (int at = wave; at < nat; at += 4) {
#pragma unroll
    for (int sp = 0; sp < 4; ++sp) {
      const int il = 16 * at + 4 * sp + g;
      const float av = (il < Bl && key < Bg) ? gl[(long long)il * Bg + key] : 0.f;     // A: key i, anchor 4 sp + g of the tile
      const float* qr = q + (long long)(row0 + (il < Bl ? il : 0)) * D;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (u < NU) {
          const int d = 16 * u + i;
          const float bv = (il < Bl && d < D) ? qr[d] : 0.f;                            // B: anchor 4 sp + g, feature 16 u + i
          dacc[u] = NCE_MFMA(av, bv, dacc[u]);
        }
      }
    }
  }
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int r = 0; r < 4; ++r) sDk[(wave * 16 + 4 * g + r) * 64 + 16 * u + i] = dacc[u][r];
  __syncthreads();
  for (int c = tid; c < 16 * 64; c += 256) {
    const int a = c >> 6, d = c & 63;
    if (j0 + a < Bg && d < D) dk[(long long)(j0 + a) * D + d] += sDk[c] + sDk[16 * 64 + c] + sDk[2 * 16 * 64 + c] + sDk[3 * 16 * 64 + c];
  }
}

// ---------------------------------------------------------------- ConR / SupCon
// block per anchor i.  prod_ij = <f_i,f_j>/t.  See contrastive.py for the masks.
__global__ __launch_bounds__(256) void ct_fwd_kernel(int mode, const float* __restrict__ fh, int B, int D, const float* __restrict__ lab_f,
                                                     const long long* __restrict__ lab_i, int C, const float* __restrict__ pred,
                                                     const float* __restrict__ wts, float w, float invt, float e, float thr,
                                                     float* __restrict__ loss, float* __restrict__ G, float* __restrict__ row_ws) {
  extern __shared__ float sm[];  // frow[D] | prod[B] | red[4]
  float* frow = sm;
  float* prod = sm + D;
  float* red = prod + B;
  const int i = blockIdx.x;
  for (int d = threadIdx.x; d < D; d += 256) frow[d] = fh[(long long)i * D + d];
  __syncthreads();
  float e_pos = 0.f, e_neg = 0.f, npos = 0.f, nneg = 0.f, nle = 0.f, spos = 0.f;
  for (int j = threadIdx.x; j < B; j += 256) {
    float s = 0.f;
    for (int d = 0; d < D; ++d) s += frow[d] * fh[(long long)j * D + d];
    s *= invt;
    prod[j] = s;
    bool pos, neg;
    float pw;
    if (mode == MMDTI_CT_REGRESS) {
      const float ld = fabsf(lab_f[i] - lab_f[j]);
      const bool le = ld <= w;
      pos = le && (j != i);
      neg = (!le) && (fabsf(pred[i] - pred[j]) <= w);
      pw = ld * (wts ? wts[i] : 1.f) * e;
      nle += le ? 1.f : 0.f;
    } else if (mode == MMDTI_CT_SINGLE) {
      const bool eq = lab_f[i] == lab_f[j];
      pos = eq && (j != i);
      neg = !eq;
      pw = wts ? wts[j] : 1.f;
    } else {
      int cnt = 0;
      for (int c = 0; c < C; ++c) cnt += (lab_i[(long long)i * C + c] == lab_i[(long long)j * C + c]) ? 1 : 0;
      const bool ge = ((float)cnt / (float)C) >= thr;
      pos = ge && (j != i);
      neg = !ge;
      pw = wts ? wts[j] : 1.f;
    }
    // exp(prod*mask): masked-out entries contribute exp(0)=1 to the positive sum (contrastive.py:35,52)
    e_pos += pos ? __expf(s) : 1.f;
    if (neg) e_neg += pw * __expf(s);
    npos += pos ? 1.f : 0.f;
    nneg += neg ? 1.f : 0.f;
    spos += pos ? s : 0.f;
    // stash the mask class in G for the second pass: 1 = pos, 2 = neg (scaled later)
    G[(long long)i * B + j] = pos ? 1.f : (neg ? 2.f : 0.f);
  }
  e_pos = block_sum(e_pos, red);
  e_neg = block_sum(e_neg, red);
  npos = block_sum(npos, red);
  nneg = block_sum(nneg, red);
  nle = block_sum(nle, red);
  spos = block_sum(spos, red);
  float denom = (mode == MMDTI_CT_REGRESS) ? nle : (npos == 0.f ? 1.f : npos);
  const float Dn = e_pos + e_neg;
  const float flag = nneg > 0.f ? 1.f : 0.f;
  const float li = flag * (npos * __logf(Dn) - spos) / denom;
  if (threadIdx.x == 0) {
    if (row_ws) row_ws[i] = li / (float)B;      // (summed in row order by ordered_sum_kernel: reproducible to the bit)
    else atomicAdd(loss, li / (float)B);
  }
  const float cg = flag / denom / (float)B;
  for (int j = threadIdx.x; j < B; j += 256) {
    const float cls = G[(long long)i * B + j];
    float g = 0.f;
    if (cls == 1.f) {
      g = cg * (npos / Dn * __expf(prod[j]) - 1.f);
    } else if (cls == 2.f) {
      float pw;
      if (mode == MMDTI_CT_REGRESS) pw = fabsf(lab_f[i] - lab_f[j]) * (wts ? wts[i] : 1.f) * e;
      else pw = wts ? wts[j] : 1.f;
      g = cg * (npos / Dn * pw * __expf(prod[j]));
    }
    G[(long long)i * B + j] = g;
  }
}

// out = x[0] + x[1] + ... in index order (one thread: the n per-row terms of a loss)
__global__ void ordered_sum_kernel(const float* __restrict__ x, int n, float* __restrict__ out) {
  float t = 0.f;
  for (int i = 0; i < n; ++i) t += x[i];
  *out = t;
}

__global__ __launch_bounds__(256) void ct_bwd_kernel(const float* __restrict__ fh, const float* __restrict__ G, int B, int D, float invt,
                                                     float* __restrict__ dfh) {
  extern __shared__ float gs[];  // [B] combined coefficient
  const int i = blockIdx.x;
  for (int j = threadIdx.x; j < B; j += 256) gs[j] = (G[(long long)i * B + j] + G[(long long)j * B + i]) * invt;
  __syncthreads();
  for (int d = threadIdx.x; d < D; d += 256) {
    float s = 0.f;
    for (int j = 0; j < B; ++j) s += gs[j] * fh[(long long)j * D + d];
    dfh[(long long)i * D + d] = s;
  }
}

// ---------------------------------------------------------------- FDS
// python-style floor division in fp32 (c10::div_floor_floating), as `(value - min)//bin_width` evaluates (fds.py:125)
__device__ __forceinline__ float div_floor_f32(float a, float b) {
  if (b == 0.f) return a / b;
  float mod = fmodf(a, b);
  float div = (a - mod) / b;
  if ((mod != 0.f) && ((b < 0.f) != (mod < 0.f))) div -= 1.f;
  float fd;
  if (div != 0.f) {
    fd = floorf(div);
    if (div - fd > 0.5f) fd += 1.f;
  } else {
    fd = copysignf(0.f, a / b);
  }
  return fd;
}

__global__ __launch_bounds__(256) void fds_bins_kernel(const float* __restrict__ labels, int n, float minv, float bw, int bs, int bn,
                                                       int* __restrict__ bins, int* __restrict__ flags) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int b = (int)div_floor_f32(labels[i] - minv, bw);
  bins[i] = b;
  if (b == bs) atomicOr(&flags[0], 1);
  if (b == bn - 1) atomicOr(&flags[1], 1);
}

// effective bucket of a sample (fds.py:134-139,167-177): first bucket takes bins <= start, last takes bins >= num-1, but
// only when some sample of the batch sits exactly in that bucket (the reference iterates torch.unique(label_bin)).
__device__ __forceinline__ int fds_eff_bucket(int b, int bs, int bn, const int* flags) {
  if (b <= bs) return (b == bs || flags[0]) ? bs : -1;                          // `label == bucket_start` branch: rows <=
  if (bs != bn - 1 && b >= bn - 1) return (b == bn - 1 || flags[1]) ? bn - 1 : -1;  // last bucket: rows >=
  if (b > bn - 1) return -1;
  return b;
}

// one wave per sample row
__global__ __launch_bounds__(256) void fds_smooth_kernel(const float* __restrict__ x, const int* __restrict__ bins, const int* __restrict__ flags,
                                                         int n, int D, int bs, int bn, const float* __restrict__ m1, const float* __restrict__ v1,
                                                         const float* __restrict__ m2, const float* __restrict__ v2, float* __restrict__ y,
                                                         float* __restrict__ scale_out) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= n) return;
  const int eb = fds_eff_bucket(bins[r], bs, bn, flags);
  bool active = eb >= 0;
  long long so = 0;
  if (active) {
    so = (long long)(eb - bs) * D;
    float s = 0.f;
    for (int d = lane; d < D; d += 64) s += v1[so + d];
    active = !(wave_sum(s) < 1e-10f);  // calibrate_mean_var early-out (utils/util.py:160-161)
  }
  for (int d = lane; d < D; d += 64) {
    float xv = x[(long long)r * D + d], sc = 1.f, o = xv;
    if (active) {
      const float a = v1[so + d];
      if (a != 0.f) {  // zero-variance columns stay untouched (util.py:162-166)
        sc = sqrtf(fminf(fmaxf(v2[so + d] / a, 0.1f), 10.f));
        o = (xv - m1[so + d]) * sc + m2[so + d];
      }
    }
    y[(long long)r * D + d] = o;
    if (scale_out) scale_out[(long long)r * D + d] = sc;
  }
}

// block per bucket, thread per feature column (strided)
__global__ __launch_bounds__(256) void fds_update_kernel(const float* __restrict__ f, const int* __restrict__ bins, const int* __restrict__ flags,
                                                         int n, int D, int bs, int bn, float factor, float* __restrict__ rmean,
                                                         float* __restrict__ rvar, float* __restrict__ tracked) {
  const int bucket = bs + blockIdx.x;
  __shared__ int s_cnt, s_exact;
  if (threadIdx.x == 0) { s_cnt = 0; s_exact = 0; }
  __syncthreads();
  int c = 0, ex = 0;
  for (int i = threadIdx.x; i < n; i += 256) {
    c += (fds_eff_bucket(bins[i], bs, bn, flags) == bucket) ? 1 : 0;
    ex += (bins[i] == bucket) ? 1 : 0;
  }
  atomicAdd(&s_cnt, c);
  atomicAdd(&s_exact, ex);
  __syncthreads();
  const int cnt = s_cnt;
  if (s_exact == 0 || cnt == 0) return;  // bucket not in torch.unique(label_bin): untouched
  // Welford, like ATen's var kernel: a constant column yields EXACTLY zero variance, which calibrate_mean_var's
  // `v1 == 0` test (utils/util.py:162) depends on.
  for (int d = threadIdx.x; d < D; d += 256) {
    float mean = 0.f, m2 = 0.f;
    int k = 0;
    for (int i = 0; i < n; ++i) {
      if (fds_eff_bucket(bins[i], bs, bn, flags) != bucket) continue;
      const float xv = f[(long long)i * D + d];
      ++k;
      const float delta = xv - mean;
      mean += delta / (float)k;
      m2 += delta * (xv - mean);
    }
    const float var = cnt > 1 ? m2 / (float)(cnt - 1) : m2 / (float)cnt;
    const long long o = (long long)blockIdx.x * D + d;
    rmean[o] = (1.f - factor) * mean + factor * rmean[o];
    rvar[o] = (1.f - factor) * var + factor * rvar[o];
  }
  if (threadIdx.x == 0) tracked[blockIdx.x] += (float)cnt;
}

// reflect-pad + conv1d across the bucket axis (fds.py:86-99)
__global__ __launch_bounds__(256) void fds_smooth_stats_kernel(const float* __restrict__ stat, int nb, int D, const float* __restrict__ win,
                                                               int ks, float* __restrict__ out) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= nb * D) return;
  const int b = t / D, d = t - b * D, half = (ks - 1) / 2;
  float s = 0.f;
  for (int k = 0; k < ks; ++k) {
    int j = b + k - half;
    if (j < 0) j = -j;
    if (j >= nb) j = 2 * (nb - 1) - j;
    s += win[k] * stat[(long long)j * D + d];
  }
  out[t] = s;
}

// ---------------------------------------------------------------- small fp32 linear (classification head)
// One LDS-tiled fp32 kernel for the three products of a small Linear kept in fp32 (mm_model.py:44-84: [B,512] -> 512 ->
// out): 32 x 32 output tiles on four waves (16 x 16 each), the products on the matrix pipe as v_mfma_f32_16x16x4_f32 -- fp32
// products, fp32 accumulation: the head keeps fp32 precision.  128-deep k slices, the NEXT slice's global loads issued before the
// current one is multiplied; every global access runs along the operand's contiguous axis.  dz = dy * tanh'(y) is formed while
// the tile is loaded.  (Rounds 1-3 multiplied on the VALU from 32-deep slices: five LDS reads per four FMAs and four exposed global
// round trips -- 55-73 us per launch on the step's serial stretch between forward and backward; now 8-12 us.)
//   MODE 0: y[r,o]   = act(sum_k x[r,k] W[o,k] + b[o])           M = rows,  N = out_f,    K = in_f
//   MODE 1: dx[r,k]  =      sum_o dz[r,o] W[o,k]                  M = rows,  N = in_f,     K = out_f
//   MODE 2: dW[o,k] +=      sum_r dz[r,o] x[r,k] ; db[o] += sum_r dz[r,o]  (column k == in_f)   M = out_f, N = in_f + 1, K = rows
constexpr int LF_KS = 128;       // k slice depth
constexpr int LF_SA = LF_KS + 1; // row stride of the A image [32 m][k]
constexpr int LF_SB = 33;        // row stride of the B image [k][32 n]
template <int MODE>
__global__ __launch_bounds__(256) void linear_f32_tile_kernel(const float* __restrict__ x, const float* __restrict__ W, const float* __restrict__ b,
                                                              const float* __restrict__ y, const float* __restrict__ dy, int rows, int in_f,
                                                              int out_f, int act, float* __restrict__ out, float* __restrict__ db) {
  __shared__ float sA[32 * LF_SA];   // [m][k]
  __shared__ float sB[LF_KS * LF_SB];   // [k][n]
  const int M = MODE == 2 ? out_f : rows, N = MODE == 0 ? out_f : (MODE == 1 ? in_f : in_f + 1), K = MODE == 0 ? in_f : (MODE == 1 ? out_f : rows);
  const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wy = wave >> 1, wx = wave & 1, li = lane & 15, lq = lane >> 4;
  auto dz = [&](int r, int o) {
    float g = dy[(long long)r * out_f + o];
    if (act == 3) {
      const float yy = y[(long long)r * out_f + o];
      g *= 1.f - yy * yy;
    }
    return g;
  };
  // element e (0..15) of this thread in the A / B images of a slice: A is 32 x 128, B is 128 x 32 -- 4096 values each, 16 per thread,
  // indexed so that consecutive threads walk the operand's contiguous axis in global memory
  float ra[16], rb[16];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int t = e * 256 + tid;
      if (MODE == 0) {          // A[m][k] = x[m, k] (k contiguous);  B[k][n] = W[n, k] (k contiguous)
        const int mm = t >> 7, kk = t & 127;
        ra[e] = (m0 + mm < M && k0 + kk < K) ? x[(long long)(m0 + mm) * in_f + k0 + kk] : 0.f;
        rb[e] = (n0 + mm < N && k0 + kk < K) ? W[(long long)(n0 + mm) * in_f + k0 + kk] : 0.f;      // (n = mm here)
      } else if (MODE == 1) {   // A[m][k] = dz(m, k) (k contiguous);  B[k][n] = W[k, n] (n contiguous)
        const int mm = t >> 7, kk = t & 127;
        ra[e] = (m0 + mm < M && k0 + kk < K) ? dz(m0 + mm, k0 + kk) : 0.f;
        const int kb = t >> 5, nn = t & 31;
        rb[e] = (k0 + kb < K && n0 + nn < N) ? W[(long long)(k0 + kb) * in_f + n0 + nn] : 0.f;
      } else {                  // A[m][k] = dz(k, m) (m contiguous);  B[k][n] = x[k, n] | 1 (n contiguous)
        const int kb = t >> 5, mm = t & 31;
        ra[e] = (k0 + kb < K && m0 + mm < M) ? dz(k0 + kb, m0 + mm) : 0.f;
        rb[e] = (k0 + kb < K && n0 + mm < N) ? (n0 + mm < in_f ? x[(long long)(k0 + kb) * in_f + n0 + mm] : 1.f) : 0.f;   // (n = mm here)
      }
    }
  };
  auto park = [&]() {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int t = e * 256 + tid;
      if (MODE == 0) {
        const int mm = t >> 7, kk = t & 127;
        sA[mm * LF_SA + kk] = ra[e];
        sB[kk * LF_SB + mm] = rb[e];
      } else if (MODE == 1) {
        sA[(t >> 7) * LF_SA + (t & 127)] = ra[e];
        sB[(t >> 5) * LF_SB + (t & 31)] = rb[e];
      } else {
        sA[(t & 31) * LF_SA + (t >> 5)] = ra[e];
        sB[(t >> 5) * LF_SB + (t & 31)] = rb[e];
      }
    }
  };
  nce_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  fetch(0);
  for (int k0 = 0; k0 < K; k0 += LF_KS) {
    __syncthreads();            // (the previous slice has been multiplied)
    park();
    __syncthreads();
    if (k0 + LF_KS < K) fetch(k0 + LF_KS);
    const int kn = min(LF_KS, K - k0);
    const float* pa = sA + (wy * 16 + li) * LF_SA + lq;
    const float* pb = sB + lq * LF_SB + wx * 16 + li;
    for (int kk = 0; kk < kn; kk += 4) acc = NCE_MFMA(pa[kk], pb[kk * LF_SB], acc);      // (slots past K hold zeros)
  }
  // accumulator: rows 4 lq + r, column li of this wave's 16 x 16 tile
  const int n = n0 + wx * 16 + li;
  if (n >= N) return;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int m = m0 + wy * 16 + 4 * lq + r;
    if (m >= M) continue;
    if (MODE == 0) {
      const float v = acc[r] + (b ? b[n] : 0.f);
      out[(long long)m * out_f + n] = act == 3 ? tanhf(v) : v;
    } else if (MODE == 1) {
      out[(long long)m * in_f + n] = acc[r];
    } else if (n < in_f) {
      out[(long long)m * in_f + n] += acc[r];     // (one lane per element: the accumulation needs no atomic)
    } else if (db) {
      db[m] += acc[r];
    }
  }
}

// ---------------------------------------------------------------- task losses
__global__ __launch_bounds__(256) void mse_kernel(const float* __restrict__ p, const float* __restrict__ t, int n, float* __restrict__ loss, float* __restrict__ dp) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) {
    const float d = p[i] - t[i];
    s += d * d;
    dp[i] = 2.f * d / (float)n;
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) *loss = s / (float)n;
}
// nn.BCEWithLogitsLoss(), mean over all B x C elements: max(x, 0) - x t + log1p(exp(-|x|)); d/dx = sigmoid(x) - t
__global__ __launch_bounds__(256) void bce_logits_kernel(const float* __restrict__ x, const float* __restrict__ t, int n, float* __restrict__ loss,
                                                         float* __restrict__ dx) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) {
    const float xi = x[i], ti = t[i];
    const float e = expf(-fabsf(xi));
    s += fmaxf(xi, 0.f) - xi * ti + log1pf(e);
    const float sig = xi >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
    dx[i] = (sig - ti) / (float)n;
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) *loss = s / (float)n;
}
__global__ __launch_bounds__(256) void ce_kernel(const float* __restrict__ lg, const long long* __restrict__ tg, int B, int C, float* __restrict__ loss,
                                                 float* __restrict__ dl) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < B; i += 256) {
    float m = -INFINITY;
    for (int c = 0; c < C; ++c) m = fmaxf(m, lg[(long long)i * C + c]);
    float se = 0.f;
    for (int c = 0; c < C; ++c) se += expf(lg[(long long)i * C + c] - m);
    const float lse = m + logf(se);
    const int y = (int)tg[i];
    s += lse - lg[(long long)i * C + y];
    for (int c = 0; c < C; ++c) dl[(long long)i * C + c] = (expf(lg[(long long)i * C + c] - lse) - (c == y ? 1.f : 0.f)) / (float)B;
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) *loss = s / (float)B;
}

}  // namespace mmdti
using namespace mmdti;

extern "C" int mmdti_seq_mean_fwd(mmdti_stream_t stream, const void* x_bf16, int B, int S, int D, int ld, float* out) {
  MMDTI_REQUIRE(x_bf16 && out && B > 0 && S > 0 && D > 0 && ld >= D, "seq_mean_fwd: bad arguments");
  if (D % 8 == 0 && ld % 8 == 0 && D >= 64 && D <= 4096 && aligned16(x_bf16)) {
    hipLaunchKernelGGL(seq_mean_fwd_vec_kernel, dim3(B, cdiv(D, 64)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x_bf16, S, D, ld, (const int*)nullptr,
                       (const int*)nullptr, out);
  } else {
    hipLaunchKernelGGL(seq_mean_fwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x_bf16, S, D, ld, out);
  }
  MMDTI_LAUNCH_CHECK();
  return MMDTI_OK;
}
extern "C" int mmdti_seq_mean_bwd(mmdti_stream_t stream, const float* dout, int B, int S, int D, int ld, void* dx_bf16, const void* aux_bf16,
                                  int ld_aux, int aux_mode) {
  MMDTI_REQUIRE(dout && dx_bf16 && B > 0 && S > 0 && D > 0 && ld >= D, "seq_mean_bwd: bad arguments");
  MMDTI_REQUIRE(aux_mode >= 0 && aux_mode <= 2 && (aux_mode == 0 || aux_bf16), "seq_mean_bwd: aux_mode %d needs aux", aux_mode);
  if (ld % 8 == 0 && aligned16(dx_bf16) && (aux_mode == 0 || (ld_aux % 8 == 0 && ld_aux >= ld && aligned16(aux_bf16)))) {
    const long long rows = (long long)B * S, chunks = rows * (ld / 8);
    hipLaunchKernelGGL(seq_mean_bwd_vec_kernel, dim3((unsigned)cdiv(chunks, 256)), dim3(256), 0, (hipStream_t)stream, dout, S, D, ld,
                       (const bf16_t*)aux_bf16, ld_aux, aux_mode, (bf16_t*)dx_bf16, rows);
  } else {
    MMDTI_REQUIRE(aux_mode == 0, "seq_mean_bwd: the aux factor needs 8-column aligned rows");
    hipLaunchKernelGGL(seq_mean_bwd_kernel, dim3(B * S), dim3(256), 0, (hipStream_t)stream, dout, S, D, ld, (bf16_t*)dx_bf16);
  }
  MMDTI_LAUNCH_CHECK();
  return MMDTI_OK;
}
extern "C" int mmdti_seq_mean_packed_fwd(mmdti_stream_t stream, const void* x_bf16, int B, int S, int D, int ld, const int* row_off,
                                         const int* n_real, float* out) {
  MMDTI_REQUIRE(x_bf16 && out && row_off && n_real && B > 0 && S > 0 && D > 0 && ld >= D, "seq_mean_packed_fwd: bad arguments");
  MMDTI_REQUIRE(D % 8 == 0 && ld % 8 == 0 && D <= 4096 && aligned16(x_bf16), "seq_mean_packed_fwd: D %% 8, ld %% 8, D <= 4096 and 16-byte alignment required");
  // (a sequence has at most S rows: S - 1 real tokens + the representative pad row, or S real tokens; fixed summation order)
  hipLaunchKernelGGL(seq_mean_fwd_vec_kernel, dim3(B, cdiv(D, 64)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x_bf16, S, D, ld, row_off, n_real, out);
  MMDTI_LAUNCH_CHECK();
  return MMDTI_OK;
}
extern "C" int mmdti_seq_mean_packed_bwd(mmdti_stream_t stream, const float* dout, int rows, int S, int D, int ld, const int* row_off,
                                         const int* n_real, const int* row_seq, void* dx_bf16, const void* aux_bf16, int ld_aux, int aux_mode) {
  MMDTI_REQUIRE(dout && dx_bf16 && row_off && n_real && row_seq && rows > 0 && S > 0 && D > 0 && ld >= D, "seq_mean_packed_bwd: bad arguments");
  MMDTI_REQUIRE(aux_mode >= 0 && aux_mode <= 2 && (aux_mode == 0 || aux_bf16), "seq_mean_packed_bwd: aux_mode %d needs aux", aux_mode);
  MMDTI_REQUIRE(ld % 8 == 0 && aligned16(dx_bf16) && (aux_mode == 0 || (ld_aux % 8 == 0 && ld_aux >= ld && aligned16(aux_bf16))),
                "seq_mean_packed_bwd: 8-column aligned rows required");
  const long long chunks = (long long)rows * (ld / 8);
  hipLaunchKernelGGL(seq_mean_packed_bwd_kernel, dim3((unsigned)cdiv(chunks, 256)), dim3(256), 0, (hipStream_t)stream, dout, S, D, ld, row_off,
                     n_real, row_seq, (const bf16_t*)aux_bf16, ld_aux, aux_mode, (bf16_t*)dx_bf16, (long long)rows);
  MMDTI_LAUNCH_CHECK();
  return MMDTI_OK;
}
extern "C" int mmdti_l2norm_fwd(mmdti_stream_t stream, const float* x, int B, int D, int ldx, float* xhat, float* inv_norm) {
  MMDTI_REQUIRE(x && xhat && inv_norm && B > 0 && D > 0 && ldx >= D, "l2norm_fwd: bad arguments");
  hipLaunchKernelGGL(l2norm_fwd_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, x, D, ldx, xhat, inv_norm);
  MMDTI_LAUNCH_CHECK();
  return MMDTI_OK;
}
extern "C" int mmdti_l2norm_bwd(mmdti_stream_t stream, const float* dxhat, const float* xhat, const float* inv_norm, int B, int D, int ldx, float* dx) {
  MMDTI_REQUIRE(dxhat && xhat && inv_norm && dx && B > 0 && D > 0 && ldx >= D, "l2norm_bwd: bad arguments");
  hipLaunchKernelGGL(l2norm_bwd_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, dxhat, xhat, inv_norm, D, ldx, dx);
  MMDTI_LAUNCH_CHECK();
  return MMDTI_OK;
}
extern "C" int mmdti_infonce_dir(mmdti_stream_t stream, const float* qh_all, const float* kh_all, int Bg, int D, int row0, int Bl,
                                 float temperature, float* loss_sum, float* dq_all, float* dk_all, float* scratch) {
  MMDTI_REQUIRE(qh_all && kh_all && loss_sum && dq_all && dk_all && scratch, "infonce_dir: null pointer");
  MMDTI_REQUIRE(Bg > 0 && D > 0 && row0 >= 0 && Bl > 0 && row0 + Bl <= Bg, "infonce_dir: bad sizes (Bg=%d row0=%d Bl=%d)", Bg, row0, Bl);
  MMDTI_REQUIRE(temperature > 0.f, "infonce_dir: temperature must be positive");
  if (D <= NCE_MAXD) {     // similarity matrix and both gradient products as fp32 MFMAs
    const int DP = (D + 3) & ~3;
    const size_t smem_m = ((size_t)16 * DP + 4 * 16 * 2 + 16 + 4 * 16 * 17 + 4 * 16 * 64) * sizeof(float);
    hipLaunchKernelGGL(infonce_rows_mfma_kernel, dim3(cdiv(Bl, 16)), dim3(256), smem_m, (hipStream_t)stream, qh_all, kh_all, Bg, D, row0, Bl,
                       1.0f / temperature, loss_sum, dq_all, scratch);
    MMDTI_LAUNCH_CHECK();
    hipLaunchKernelGGL(infonce_cols_mfma_kernel, dim3(cdiv(Bg, 16)), dim3(256), 0, (hipStream_t)stream, qh_all, scratch, Bg, D, row0, Bl, dk_all, loss_sum);
    MMDTI_LAUNCH_CHECK();
    return MMDTI_OK;
  }
  const size_t smem = ((size_t)D + Bg + 4) * sizeof(float);
  MMDTI_REQUIRE(smem <= 64 * 1024, "infonce_dir: global batch %d too large for the LDS logits row", Bg);
  hipLaunchKernelGGL(infonce_rows_kernel, dim3(Bl), dim3(256), smem, (hipStream_t)stream, qh_all, kh_all, Bg, D, row0,
                     1.0f / temperature, loss_sum, dq_all, scratch);
  MMDTI_LAUNCH_CHECK();
  hipLaunchKernelGGL(infonce_cols_kernel, dim3(Bg), dim3(64), 0, (hipStream_t)stream, qh_all, scratch, Bg, D, row0, Bl, dk_all, loss_sum);
  MMDTI_LAUNCH_CHECK();
  return MMDTI_OK;
}

extern "C" int mmdti_ct_loss_fwd(mmdti_stream_t stream, int mode, const float* fhat, int B, int D, const float* labels_f,
                                 const long long* labels_i, int C, const float* pred, const float* weights, float w, float t,
                                 float e, float coef, float* loss, float* G, float* row_ws) {
  MMDTI_REQUIRE(mode >= 0 && mode <= 2, "ct_loss_fwd: bad mode %d", mode);
  MMDTI_REQUIRE(fhat && loss && G && B > 0 && D > 0 && t > 0.f, "ct_loss_fwd: bad arguments");
  MMDTI_REQUIRE(mode == MMDTI_CT_MULTI ? (labels_i && C > 0) : (labels_f != nullptr), "ct_loss_fwd: labels missing");
  MMDTI_REQUIRE(mode != MMDTI_CT_REGRESS || pred, "ct_loss_fwd: regress needs predictions");
  const size_t smem = ((size_t)D + B + 4) * sizeof(float);
  MMDTI_REQUIRE(smem <= 64 * 1024, "ct_loss_fwd: B+D too large for LDS");
  hipStream_t s = (hipStream_t)stream;
  // (row_ws: the per-row terms are stored and summed in row order by one thread; without it they meet in an fp32 atomic)
  if (!row_ws && hipMemsetAsync(loss, 0, sizeof(float), s) != hipSuccess) { set_error("ct_loss_fwd: memset failed"); return MMDTI_ERR_LAUNCH; }
  const float thr = mode == MMDTI_CT_MULTI ? (float)((double)coef / (double)C) : 0.f;
  hipLaunchKernelGGL(ct_fwd_kernel, dim3(B), dim3(256), smem, s, mode, fhat, B, D, labels_f, labels_i, C, pred, weights, w,
                     1.0f / t, e, thr, loss, G, row_ws);
  if (row_ws) hipLaunchKernelGGL(ordered_sum_kernel, dim3(1), dim3(1), 0, s, row_ws, B, loss);
  MMDTI_LAUNCH_CHECK();
  return MMDTI_OK;
}
extern "C" int mmdti_ct_loss_bwd(mmdti_stream_t stream, const float* fhat, const float* G, int B, int D, float t, float* dfhat) {
  MMDTI_REQUIRE(fhat && G && dfhat && B > 0 && D > 0 && t > 0.f, "ct_loss_bwd: bad arguments");
  MMDTI_REQUIRE((size_t)B * sizeof(float) <= 64 * 1024, "ct_loss_bwd: B too large");
  hipLaunchKernelGGL(ct_bwd_kernel, dim3(B), dim3(256), (size_t)B * sizeof(float), (hipStream_t)stream, fhat, G, B, D, 1.0f / t, dfhat);
  MMDTI_LAUNCH_CHECK();
  return MMDTI_OK;
}

extern "C" int mmdti_fds_bins(mmdti_stream_t stream, const float* labels, int n, float min_value, float bin_width, int bucket_start,
                              int bucket_num, int* bins, int* flags) {
  MMDTI_REQUIRE(labels && bins && flags && n > 0 && bucket_num > bucket_start && bucket_start >= 0, "fds_bins: bad arguments");
  hipLaunchKernelGGL(fds_bins_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, labels, n, min_value, bin_width,
                     bucket_start, bucket_num, bins, flags);
  MMDTI_LAUNCH_CHECK();
  return MMDTI_OK;
}
extern "C" int mmdti_fds_smooth(mmdti_stream_t stream, const float* x, const int* bins, const int* flags, int n, int D, int bucket_start,
                                int bucket_num, const float* m1, const float* v1, const float* m2, const float* v2, float* y,
                                float* scale_out) {
  MMDTI_REQUIRE(x && bins && flags && m1 && v1 && m2 && v2 && y && n > 0 && D > 0, "fds_smooth: bad arguments");
  hipLaunchKernelGGL(fds_smooth_kernel, dim3(cdiv(n, 4)), dim3(256), 0, (hipStream_t)stream, x, bins, flags, n, D, bucket_start,
                     bucket_num, m1, v1, m2, v2, y, scale_out);
  MMDTI_LAUNCH_CHECK();
  return MMDTI_OK;
}
extern "C" int mmdti_fds_update_stats(mmdti_stream_t stream, const float* feats, const int* bins, const int* flags, int n, int D,
                                      int bucket_start, int bucket_num, float factor, float* running_mean, float* running_var,
                                      float* num_samples_tracked) {
  MMDTI_REQUIRE(feats && bins && flags && running_mean && running_var && num_samples_tracked && n > 0 && D > 0, "fds_update_stats: bad arguments");
  MMDTI_REQUIRE(bucket_num > bucket_start, "fds_update_stats: empty bucket range");
  hipLaunchKernelGGL(fds_update_kernel, dim3(bucket_num - bucket_start), dim3(256), 0, (hipStream_t)stream, feats, bins, flags, n, D,
                     bucket_start, bucket_num, factor, running_mean, running_var, num_samples_tracked);
  MMDTI_LAUNCH_CHECK();
  return MMDTI_OK;
}
extern "C" int mmdti_fds_smooth_stats(mmdti_stream_t stream, const float* stat, int nb, int D, const float* window, int ks, float* out) {
  MMDTI_REQUIRE(stat && window && out && nb > 0 && D > 0 && ks > 0 && (ks - 1) / 2 < nb, "fds_smooth_stats: bad arguments");
  hipLaunchKernelGGL(fds_smooth_stats_kernel, dim3(cdiv((long long)nb * D, 256)), dim3(256), 0, (hipStream_t)stream, stat, nb, D, window, ks, out);
  MMDTI_LAUNCH_CHECK();
  return MMDTI_OK;
}

extern "C" int mmdti_linear_f32_fwd(mmdti_stream_t stream, const float* x, const float* W, const float* b, int rows, int in_f, int out_f,
                                    int act, float* y) {
  MMDTI_REQUIRE(x && W && y && rows > 0 && in_f > 0 && out_f > 0 && (act == 0 || act == 3), "linear_f32_fwd: bad arguments");
  hipLaunchKernelGGL(linear_f32_tile_kernel<0>, dim3(cdiv(out_f, 32), cdiv(rows, 32)), dim3(256), 0, (hipStream_t)stream, x, W, b,
                     (const float*)nullptr, (const float*)nullptr, rows, in_f, out_f, act, y, (float*)nullptr);
  MMDTI_LAUNCH_CHECK();
  return MMDTI_OK;
}
extern "C" int mmdti_linear_f32_bwd(mmdti_stream_t stream, const float* x, const float* W, const float* y, const float* dy, int rows,
                                    int in_f, int out_f, int act, float* dx, float* dW, float* db) {
  MMDTI_REQUIRE(x && W && dy && rows > 0 && in_f > 0 && out_f > 0 && (act == 0 || (act == 3 && y)), "linear_f32_bwd: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  if (dx) hipLaunchKernelGGL(linear_f32_tile_kernel<1>, dim3(cdiv(in_f, 32), cdiv(rows, 32)), dim3(256), 0, s, x, W, (const float*)nullptr, y, dy,
                             rows, in_f, out_f, act, dx, (float*)nullptr);
  if (dW) hipLaunchKernelGGL(linear_f32_tile_kernel<2>, dim3(cdiv(in_f + 1, 32), cdiv(out_f, 32)), dim3(256), 0, s, x, W, (const float*)nullptr, y, dy,
                             rows, in_f, out_f, act, dW, db);
  MMDTI_LAUNCH_CHECK();
  return MMDTI_OK;
}
extern "C" int mmdti_mse_loss(mmdti_stream_t stream, const float* pred, const float* target, int n, float* loss, float* dpred) {
  MMDTI_REQUIRE(pred && target && loss && dpred && n > 0, "mse_loss: bad arguments");
  hipLaunchKernelGGL(mse_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, pred, target, n, loss, dpred);
  MMDTI_LAUNCH_CHECK();
  return MMDTI_OK;
}
extern "C" int mmdti_bce_logits_loss(mmdti_stream_t stream, const float* logits, const float* target, int n, float* loss, float* dlogits) {
  MMDTI_REQUIRE(logits && target && loss && dlogits && n > 0, "bce_logits_loss: bad arguments");
  hipLaunchKernelGGL(bce_logits_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logits, target, n, loss, dlogits);
  MMDTI_LAUNCH_CHECK();
  return MMDTI_OK;
}
extern "C" int mmdti_ce_loss(mmdti_stream_t stream, const float* logits, const long long* target, int B, int C, float* loss, float* dlogits) {
  MMDTI_REQUIRE(logits && target && loss && dlogits && B > 0 && C > 0, "ce_loss: bad arguments");
  hipLaunchKernelGGL(ce_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logits, target, B, C, loss, dlogits);
  MMDTI_LAUNCH_CHECK();
  return MMDTI_OK;
}
