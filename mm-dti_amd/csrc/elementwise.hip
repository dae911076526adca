// Bandwidth-bound helpers of the fine-tune step: casts, dropout, column sums, embeddings, RoBERTa position ids,
// row softmax over materialised attention scores, masked pooling, GELU, sum of squares, Adam.
// All are simple streams; 16-byte accesses per lane wherever the layout allows (cdna guide G13).
#include "common.h"
#include <stdarg.h>
#include <stdio.h>

namespace mmdti {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

static inline int grid_for(long long n, int per_block) {
  long long b = (n + per_block - 1) / per_block;
  if (b > 256 * 16) b = 256 * 16;
  if (b < 1) b = 1;
  return (int)b;
}

// ---------------------------------------------------------------- casts / dropout / axpy
__device__ __forceinline__ bf16_t f2h16(float f) { return f2h_sat(f); }

// F16: the 16-bit output is fp16 (the fp16 forward-operand mode), else bf16
template <bool F16>
__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ y,
                                                            long long n4, long long n, uint32_t thresh, float dscale,
                                                            uint64_t seed, uint32_t site) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    float4 v = reinterpret_cast<const float4*>(x)[i];
    float o[4] = {v.x, v.y, v.z, v.w};
    if (thresh) {
      Rand4 r = philox4(seed, site, (uint64_t)i);
      uint32_t rw[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = rw[e] >= thresh ? o[e] * dscale : 0.f;
    }
    uint2 pk;
    if (F16) {
      pk.x = (uint32_t)f2h16(o[0]) | ((uint32_t)f2h16(o[1]) << 16);
      pk.y = (uint32_t)f2h16(o[2]) | ((uint32_t)f2h16(o[3]) << 16);
    } else {
      pk.x = (uint32_t)f2bf(o[0]) | ((uint32_t)f2bf(o[1]) << 16);
      pk.y = (uint32_t)f2bf(o[2]) | ((uint32_t)f2bf(o[3]) << 16);
    }
    reinterpret_cast<uint2*>(y)[i] = pk;
  }
  // tail (n % 4)
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    long long i = (n & ~3LL) + threadIdx.x;
    float o = x[i];
    if (thresh) o = dropout_keep(seed, site, (uint64_t)i, thresh) ? o * dscale : 0.f;
    y[i] = F16 ? f2h16(o) : f2bf(o);
  }
}

// fp16 -> bf16 (round to nearest even), rows of `cols` elements (cols % 8 == 0) with a source row stride: the backward GEMMs of
// the fp16 forward-operand mode take the saved forward activations as bf16
__global__ __launch_bounds__(256) void cast_f16_bf16_kernel(const bf16_t* __restrict__ x, int cols8, int ldx, bf16_t* __restrict__ y, long long n8) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
    const long long row = i / cols8;
    const int c = (int)(i - row * cols8) * 8;
    const uint4 u = *reinterpret_cast<const uint4*>(x + row * ldx + c);
    const uint32_t w[4] = {u.x, u.y, u.z, u.w};
    uint32_t o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float lo = (float)__builtin_bit_cast(_Float16, (uint16_t)(w[e] & 0xffffu)), hi = (float)__builtin_bit_cast(_Float16, (uint16_t)(w[e] >> 16));
      o[e] = (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
    }
    *reinterpret_cast<uint4*>(y + row * (long long)cols8 * 8 + c) = make_uint4(o[0], o[1], o[2], o[3]);
  }
}

__global__ __launch_bounds__(256) void cast_bf16_f32_kernel(const bf16_t* __restrict__ x, float* __restrict__ y, long long n) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) y[i] = bf2f(x[i]);
}

__global__ __launch_bounds__(256) void dropout_f32_kernel(const float* __restrict__ x, float* __restrict__ y, long long n,
                                                          uint32_t thresh, float dscale, uint64_t seed, uint32_t site) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    float o = x[i];
    if (thresh) o = dropout_keep(seed, site, (uint64_t)i, thresh) ? o * dscale : 0.f;
    y[i] = o;
  }
}

__global__ __launch_bounds__(256) void axpy_kernel(const float* __restrict__ x, float* __restrict__ y, long long n, float a) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) y[i] += a * x[i];
}

__global__ __launch_bounds__(256) void gelu_bf16_kernel(const bf16_t* __restrict__ u, bf16_t* __restrict__ y, long long n) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
    y[i] = f2bf(gelu_erf(bf2f(u[i])));
}

// ---------------------------------------------------------------- column sum (bias gradients)
// block = 256 threads: 32 column-chunks(8 cols each, 16 B) x 8 row lanes; grid.x over column groups of 256, grid.y rows
__global__ __launch_bounds__(256) void colsum_bf16_kernel(const bf16_t* __restrict__ x, int rows, int cols, int ld,
                                                          float* __restrict__ out) {
  __shared__ float red[8][256 + 8];
  const int cc = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int c0 = blockIdx.x * 256 + cc * 8;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (c0 < cols) {
    const int rstep = gridDim.y * 8;
    int r = blockIdx.y * 8 + rl;
    if (c0 + 8 <= cols) {
      // 4 rows per iteration: four independent 16-byte loads in flight per lane
      for (; r + 3 * rstep < rows; r += 4 * rstep) {
        uint4 u[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) u[k] = *reinterpret_cast<const uint4*>(x + (long long)(r + k * rstep) * ld + c0);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const uint32_t w[4] = {u[k].x, u[k].y, u[k].z, u[k].w};
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            acc[2 * i] += __uint_as_float(w[i] << 16);
            acc[2 * i + 1] += __uint_as_float(w[i] & 0xffff0000u);
          }
        }
      }
      for (; r < rows; r += rstep) {
        const uint4 u = *reinterpret_cast<const uint4*>(x + (long long)r * ld + c0);
        const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          acc[2 * i] += __uint_as_float(w[i] << 16);
          acc[2 * i + 1] += __uint_as_float(w[i] & 0xffff0000u);
        }
      }
    } else {
      for (; r < rows; r += rstep) {
        const bf16_t* p = x + (long long)r * ld + c0;
        for (int i = 0; i < cols - c0; ++i) acc[i] += bf2f(p[i]);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) red[rl][cc * 8 + i] = acc[i];
  __syncthreads();
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c < cols) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) s += red[r][threadIdx.x];
    atomicAdd(out + c, s);
  }
}

// ---------------------------------------------------------------- embeddings
__global__ __launch_bounds__(256) void embedding_fwd_kernel(const long long* __restrict__ ids, const float* __restrict__ table,
                                                            long long n, int D4, int vocab, float* __restrict__ out, int accumulate) {
  const long long total = n * D4;
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
    const long long i = t / D4;
    const int c = (int)(t - i * D4);
    long long id = ids[i];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    float4 v = reinterpret_cast<const float4*>(table)[id * D4 + c];
    if (accumulate) {
      float4 o = reinterpret_cast<float4*>(out)[t];
      v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
    }
    reinterpret_cast<float4*>(out)[t] = v;
  }
}

__global__ __launch_bounds__(256) void embedding_bwd_kernel(const long long* __restrict__ ids, const float* __restrict__ dout,
                                                            long long n, int D, int vocab, long long padding_idx,
                                                            float* __restrict__ dtable) {
  const long long total = n * D;
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
    const long long i = t / D;
    const int c = (int)(t - i * D);
    long long id = ids[i];
    if (id == padding_idx || id < 0 || id >= vocab) continue;
    atomicAdd(dtable + id * D + c, dout[t]);
  }
}

// word + position + token-type embeddings summed in one pass (HF RobertaEmbeddings.forward, modeling_roberta.py:75-122): one
// 16-byte store per element instead of a store and two read-modify-write passes.  ids_c == null: row 0 of table_c (token type 0).
__global__ __launch_bounds__(256) void embedding_fwd3_kernel(const long long* __restrict__ ids_a, const float* __restrict__ ta, int va,
                                                             const long long* __restrict__ ids_b, const float* __restrict__ tb, int vb,
                                                             const long long* __restrict__ ids_c, const float* __restrict__ tc, int vc,
                                                             long long n, int D4, float* __restrict__ out) {
  const long long total = n * D4;
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
    const long long i = t / D4;
    const int c = (int)(t - i * D4);
    long long ia = ids_a[i], ib = ids_b[i], ic = ids_c ? ids_c[i] : 0;
    ia = ia < 0 ? 0 : (ia >= va ? va - 1 : ia);
    ib = ib < 0 ? 0 : (ib >= vb ? vb - 1 : ib);
    ic = ic < 0 ? 0 : (ic >= vc ? vc - 1 : ic);
    const float4 a = reinterpret_cast<const float4*>(ta)[ia * D4 + c], b = reinterpret_cast<const float4*>(tb)[ib * D4 + c],
                 d = reinterpret_cast<const float4*>(tc)[ic * D4 + c];
    reinterpret_cast<float4*>(out)[t] = make_float4(a.x + b.x + d.x, a.y + b.y + d.y, a.z + b.z + d.z, a.w + b.w + d.w);   // (a + b) + c, the reference's order
  }
}

// one-hot rows (bf16) for the embedding backward GEMM: thread per 8-column chunk
__global__ __launch_bounds__(256) void onehot_bf16_kernel(const long long* __restrict__ ids, long long n, int vocab, int ld8,
                                                          long long padding_idx, bf16_t* __restrict__ out) {
  const long long total = n * ld8;
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
    const long long i = t / ld8;
    const int c0 = (int)(t - i * ld8) * 8;
    const long long id = ids[i];
    uint32_t w[4] = {0u, 0u, 0u, 0u};
    if (id != padding_idx && id >= 0 && id < vocab && id >= c0 && id < c0 + 8) {
      const int e = (int)(id - c0);
      w[e >> 1] = (e & 1) ? 0x3F800000u : 0x00003F80u;   // bf16(1.0) = 0x3F80
    }
    reinterpret_cast<uint4*>(out)[t] = make_uint4(w[0], w[1], w[2], w[3]);
  }
}

// one wave per sequence: inclusive scan of (ids != pad) across the row in chunks of 64
__global__ __launch_bounds__(64) void roberta_posids_kernel(const long long* __restrict__ ids, int L, long long pad,
                                                            long long* __restrict__ out) {
  const int b = blockIdx.x, lane = threadIdx.x;
  int carry = 0;
  for (int l0 = 0; l0 < L; l0 += 64) {
    const int l = l0 + lane;
    const int m = (l < L && ids[(long long)b * L + l] != pad) ? 1 : 0;
    int s = m;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      int t = __shfl_up(s, o, 64);
      if (lane >= o) s += t;
    }
    if (l < L) out[(long long)b * L + l] = (long long)((carry + s) * m) + pad;
    carry += __shfl(s, 63, 64);
  }
}

// ---------------------------------------------------------------- row softmax over materialised scores
// one wave per row, 4 consecutive keys per lane (16-byte score loads, 8-byte bf16 stores); ld % 8 == 0, ld <= 1024.
// Dropout element index = row*ld + key (ld-based so that a lane's 4 keys share one RNG call).
template <int NV>
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const float* __restrict__ s, const float* __restrict__ key_add,
                                                          bf16_t* __restrict__ p_out, bf16_t* __restrict__ pd_out,
                                                          long long rows, int heads, int Lq, int Lk, int ld, uint32_t thresh,
                                                          float dscale, uint64_t seed, uint32_t site) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int b = (int)(row / ((long long)heads * Lq));
  const float* sr = s + row * ld;
  const float* ka = key_add ? key_add + (long long)b * Lk : nullptr;
  float v[NV][4];
  float m = -INFINITY;
#pragma unroll
  for (int c = 0; c < NV; ++c) {
    const int j = (c * 64 + lane) * 4;
    float4 t = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    if (j < ld) t = *reinterpret_cast<const float4*>(sr + j);
    const float tv[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[c][e] = (j + e < Lk) ? tv[e] + (ka ? ka[j + e] : 0.f) : -INFINITY;
      m = fmaxf(m, v[c][e]);
    }
  }
  m = wave_max(m);
  float sum = 0.f;
#pragma unroll
  for (int c = 0; c < NV; ++c)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[c][e] = __expf(v[c][e] - m);   // exp(-inf) = 0 in the pad columns
      sum += v[c][e];
    }
  const float inv = 1.0f / wave_sum(sum);
#pragma unroll
  for (int c = 0; c < NV; ++c) {
    const int j = (c * 64 + lane) * 4;
    if (j >= ld) continue;
    float p[4] = {v[c][0] * inv, v[c][1] * inv, v[c][2] * inv, v[c][3] * inv};
    uint2 pk;
    pk.x = (uint32_t)f2bf(p[0]) | ((uint32_t)f2bf(p[1]) << 16);
    pk.y = (uint32_t)f2bf(p[2]) | ((uint32_t)f2bf(p[3]) << 16);
    *reinterpret_cast<uint2*>(p_out + row * ld + j) = pk;
    if (pd_out != p_out) {
      if (thresh) {
        const Rand4 r = philox4(seed, site, (uint64_t)(row * ld + j) >> 2);
        p[0] = r.x >= thresh ? p[0] * dscale : 0.f; p[1] = r.y >= thresh ? p[1] * dscale : 0.f;
        p[2] = r.z >= thresh ? p[2] * dscale : 0.f; p[3] = r.w >= thresh ? p[3] * dscale : 0.f;
      }
      pk.x = (uint32_t)f2bf(p[0]) | ((uint32_t)f2bf(p[1]) << 16);
      pk.y = (uint32_t)f2bf(p[2]) | ((uint32_t)f2bf(p[3]) << 16);
      *reinterpret_cast<uint2*>(pd_out + row * ld + j) = pk;
    }
  }
}

template <int NV>
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const bf16_t* __restrict__ p_in, const float* __restrict__ dp,
                                                          bf16_t* __restrict__ ds, long long rows, int Lk, int ld, float scale,
                                                          uint32_t thresh, float dscale, uint64_t seed, uint32_t site) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float p[NV][4], d[NV][4];
  float dl = 0.f;
#pragma unroll
  for (int c = 0; c < NV; ++c) {
    const int j = (c * 64 + lane) * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e) p[c][e] = d[c][e] = 0.f;
    if (j < ld) {
      const uint2 pk = *reinterpret_cast<const uint2*>(p_in + row * ld + j);
      const float4 t = *reinterpret_cast<const float4*>(dp + row * ld + j);
      p[c][0] = __uint_as_float(pk.x << 16); p[c][1] = __uint_as_float(pk.x & 0xffff0000u);
      p[c][2] = __uint_as_float(pk.y << 16); p[c][3] = __uint_as_float(pk.y & 0xffff0000u);
      d[c][0] = t.x; d[c][1] = t.y; d[c][2] = t.z; d[c][3] = t.w;
      if (thresh) {
        const Rand4 r = philox4(seed, site, (uint64_t)(row * ld + j) >> 2);
        d[c][0] = r.x >= thresh ? d[c][0] * dscale : 0.f; d[c][1] = r.y >= thresh ? d[c][1] * dscale : 0.f;
        d[c][2] = r.z >= thresh ? d[c][2] * dscale : 0.f; d[c][3] = r.w >= thresh ? d[c][3] * dscale : 0.f;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (j + e >= Lk) p[c][e] = d[c][e] = 0.f;   // pad columns of dp are not data
        dl += p[c][e] * d[c][e];
      }
    }
  }
  dl = wave_sum(dl);
#pragma unroll
  for (int c = 0; c < NV; ++c) {
    const int j = (c * 64 + lane) * 4;
    if (j >= ld) continue;
    float o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = scale * p[c][e] * (d[c][e] - dl);
    uint2 pk;
    pk.x = (uint32_t)f2bf(o[0]) | ((uint32_t)f2bf(o[1]) << 16);
    pk.y = (uint32_t)f2bf(o[2]) | ((uint32_t)f2bf(o[3]) << 16);
    *reinterpret_cast<uint2*>(ds + row * ld + j) = pk;
  }
}

// ---------------------------------------------------------------- masked pooling (mm_model.py:572-576)
// pooled[b] = mean over the unmasked rows of [a[b]; t[b]].  grid (B, D/64): a block owns 64 columns (16 float4 lanes) and
// spreads the Na+Nt rows over 16 row groups -- 8x the workgroups of a block-per-molecule layout, 16-byte loads.
__device__ __forceinline__ int masked_pool_count(const unsigned char* ma, const unsigned char* mt, int Na, int Nt, int* cnt) {
  if (threadIdx.x == 0) *cnt = 0;
  __syncthreads();
  int c = 0;
  for (int i = threadIdx.x; i < Na + Nt; i += 256) c += (i < Na ? ma[i] : mt[i - Na]) ? 1 : 0;
  if (c) atomicAdd(cnt, c);
  __syncthreads();
  return *cnt;
}

__global__ __launch_bounds__(256) void masked_pool_fwd_kernel(const float* __restrict__ a, const float* __restrict__ t,
                                                              const unsigned char* __restrict__ ma, const unsigned char* __restrict__ mt,
                                                              int Na, int Nt, int D, float* __restrict__ pooled) {
  const int b = blockIdx.x;
  __shared__ int cnt;
  __shared__ float4 red[16][16];
  const int n = masked_pool_count(ma + (long long)b * Na, mt + (long long)b * Nt, Na, Nt, &cnt);
  const int cl = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const int col = blockIdx.y * 64 + cl * 4;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (col < D) {
    for (int i = rg; i < Na + Nt; i += 16) {
      const bool in_a = i < Na;
      const bool on = in_a ? ma[(long long)b * Na + i] : mt[(long long)b * Nt + (i - Na)];
      if (!on) continue;
      const float* src = in_a ? a + ((long long)b * Na + i) * D : t + ((long long)b * Nt + (i - Na)) * D;
      const float4 v = *reinterpret_cast<const float4*>(src + col);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  }
  red[rg][cl] = s;
  __syncthreads();
  if (rg == 0 && col < D) {
#pragma unroll
    for (int r = 1; r < 16; ++r) {
      const float4 v = red[r][cl];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    const float inv = 1.0f / (float)n;
    *reinterpret_cast<float4*>(pooled + (long long)b * D + col) = make_float4(s.x * inv, s.y * inv, s.z * inv, s.w * inv);
  }
}

// grid (B, 8): block (b, y) writes rows y, y+8, ... of da / dt, one float4 per thread per step
__global__ __launch_bounds__(256) void masked_pool_bwd_kernel(const float* __restrict__ dp, const unsigned char* __restrict__ ma,
                                                              const unsigned char* __restrict__ mt, int Na, int Nt, int D,
                                                              float* __restrict__ da, float* __restrict__ dt) {
  const int b = blockIdx.x;
  __shared__ int cnt;
  const float inv = 1.0f / (float)masked_pool_count(ma + (long long)b * Na, mt + (long long)b * Nt, Na, Nt, &cnt);
  const int nv = D >> 2;
  for (int i = blockIdx.y; i < Na + Nt; i += gridDim.y) {
    const bool in_a = i < Na;
    const bool on = in_a ? ma[(long long)b * Na + i] : mt[(long long)b * Nt + (i - Na)];
    float* dst = in_a ? da + ((long long)b * Na + i) * D : dt + ((long long)b * Nt + (i - Na)) * D;
    for (int c = threadIdx.x; c < nv; c += 256) {
      float4 g = *reinterpret_cast<const float4*>(dp + (long long)b * D + c * 4);
      g = on ? make_float4(g.x * inv, g.y * inv, g.z * inv, g.w * inv) : make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<float4*>(dst + c * 4) = g;
    }
  }
}

// Packed token rows (see mmdti_masked_pool_packed_fwd): only a sequence's real rows enter; no masks to read.
__global__ __launch_bounds__(256) void masked_pool_packed_fwd_kernel(const float* __restrict__ a, const float* __restrict__ t,
                                                                     const int* __restrict__ a_off, const int* __restrict__ a_cnt,
                                                                     const int* __restrict__ t_off, const int* __restrict__ t_cnt, int D,
                                                                     float* __restrict__ pooled) {
  const int b = blockIdx.x;
  __shared__ float4 red[16][16];
  const int na = a_cnt[b], nt = t_cnt[b];
  const float* pa = a + (long long)a_off[b] * D;
  const float* pt = t + (long long)t_off[b] * D;
  const int cl = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const int col = blockIdx.y * 64 + cl * 4;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (col < D) {
    for (int i = rg; i < na + nt; i += 16) {
      const float* src = i < na ? pa + (long long)i * D : pt + (long long)(i - na) * D;
      const float4 v = *reinterpret_cast<const float4*>(src + col);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  }
  red[rg][cl] = s;
  __syncthreads();
  if (rg == 0 && col < D) {
#pragma unroll
    for (int r = 1; r < 16; ++r) {
      const float4 v = red[r][cl];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    const float inv = 1.0f / (float)(na + nt);
    *reinterpret_cast<float4*>(pooled + (long long)b * D + col) = make_float4(s.x * inv, s.y * inv, s.z * inv, s.w * inv);
  }
}
// one float4 per thread: rows of da (first rows_a rows of the index space), then rows of dt
__global__ __launch_bounds__(256) void masked_pool_packed_bwd_kernel(const float* __restrict__ dp, const int* __restrict__ a_off,
                                                                     const int* __restrict__ a_cnt, const int* __restrict__ a_seq, int rows_a,
                                                                     const int* __restrict__ t_off, const int* __restrict__ t_cnt,
                                                                     const int* __restrict__ t_seq, int rows_t, int D, float* __restrict__ da,
                                                                     float* __restrict__ dt) {
  const int nv = D >> 2;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long row = idx / nv;
  if (row >= (long long)rows_a + rows_t) return;
  const int c = (int)(idx - row * nv) * 4;
  const bool in_a = row < rows_a;
  const int r = in_a ? (int)row : (int)(row - rows_a);
  const int b = in_a ? a_seq[r] : t_seq[r];
  const bool on = in_a ? (r - a_off[b] < a_cnt[b]) : (r - t_off[b] < t_cnt[b]);
  const float inv = on ? 1.0f / (float)(a_cnt[b] + t_cnt[b]) : 0.f;
  const float4 g = *reinterpret_cast<const float4*>(dp + (long long)b * D + c);
  float* dst = (in_a ? da : dt) + (long long)r * D + c;
  *reinterpret_cast<float4*>(dst) = make_float4(g.x * inv, g.y * inv, g.z * inv, g.w * inv);
}

// ---------------------------------------------------------------- sum of squares / Adam
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, long long n, float* __restrict__ out) {
  float s = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) s += g[i] * g[i];
  s = wave_sum(s);
  __shared__ float red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}
// The reproducible form (a workspace is given): 16-byte loads, one partial per workgroup, and the partials folded in a FIXED order by
// sumsq_fold_kernel -- the gradient norm, and with it the clip coefficient of every Adam step, is the same to the bit from run to run
// (the atomic form above differs in the last bit, and a training trajectory with it).
constexpr int SUMSQ_WGS = 2048;
__global__ __launch_bounds__(256) void sumsq_part_kernel(const float* __restrict__ g, long long n, float* __restrict__ part) {
  const long long n4 = n >> 2;
  const float4* g4 = reinterpret_cast<const float4*>(g);
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const float4 v = g4[i];
    a.x += v.x * v.x; a.y += v.y * v.y; a.z += v.z * v.z; a.w += v.w * v.w;
  }
  float s = (a.x + a.y) + (a.z + a.w);
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const float v = g[(n4 << 2) + threadIdx.x]; s += v * v; }
  s = wave_sum(s);
  __shared__ float red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void sumsq_fold_kernel(const float* __restrict__ part, int np, float* __restrict__ out) {
  __shared__ float red[256];
  float s = 0.f;
  for (int i = threadIdx.x; i < np; i += 256) s += part[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) *out += red[0];
}

// Step state kept on the device so that a whole training step can be replayed from a captured HIP graph (host-side
// counters would be frozen into the graph): st[0] = optimizer steps taken, st[1] = learning rate of THIS step (HF linear
// warm-up / decay, transformers.get_linear_schedule_with_warmup as tasks/trainer.py:161-162 uses it), st[2] = 1 - beta1^t,
// st[3] = sqrt(1 - beta2^t); salt = the per-step dropout salt (mmdti_seed_salt_pull copies it into every kernel library).
__global__ void step_state_advance_kernel(float* __restrict__ st, unsigned long long* __restrict__ salt, float base_lr, int warmup, int total,
                                          float b1, float b2) {
  const int k = (int)st[0];                      // scheduler steps already taken: the schedule's lambda(k) is this step's rate
  const float lam = k < warmup ? (float)k / (float)max(1, warmup) : fmaxf(0.f, (float)(total - k) / (float)max(1, total - warmup));
  const float t = (float)(k + 1);
  st[0] = t;
  st[1] = base_lr * lam;
  st[2] = 1.f - powf(b1, t);
  st[3] = sqrtf(1.f - powf(b2, t));
  unsigned long long x = *salt + 0x9E3779B97F4A7C15ull;       // splitmix64: a fresh, well-mixed word per step
  unsigned long long z = x;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  *salt = x;
  salt[1] = z ^ (z >> 31);
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, bf16_t* __restrict__ pb, bf16_t* __restrict__ ph, long long n, float lr, float b1,
                                                   float b2, float eps, float wd, float bc1, float bc2_sqrt,
                                                   const float* __restrict__ gscale, const float* __restrict__ state) {
  const float gs = gscale ? *gscale : 1.0f;
  if (state) {               // device-resident schedule (graph replay): this step's rate and bias corrections
    lr = state[1];
    bc1 = state[2];
    bc2_sqrt = state[3];
  }
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    float gi = g[i] * gs;
    float pi = p[i];
    if (wd != 0.f) gi += wd * pi;
    float mi = b1 * m[i] + (1.f - b1) * gi;
    float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    pi -= (lr / bc1) * mi / denom;
    p[i] = pi;
    if (pb) pb[i] = f2bf(pi);
    if (ph) ph[i] = f2h_sat(pi);      // the fp16 shadow the forward GEMMs read (fp16 forward-operand mode)
  }
}

// ---------------------------------------------------------------- hardware probe: ds_read_b64_tr_b16 semantics
__global__ __launch_bounds__(64) void probe_tr_kernel(int stride, unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short img[64 * 64];
  for (int i = threadIdx.x; i < 64 * 64; i += 64) img[i] = (unsigned short)i;
  __syncthreads();
  const int lane = threadIdx.x;
  const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
  // group g reads the 4x16 block starting at row 4*g, column 0: lane 4q+p supplies &img[(4g+q)*stride + 4p]
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  const unsigned short* src = &img[(4 * g + q) * stride + 4 * pp];
  s16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)src);
  for (int i = 0; i < 4; ++i) out[lane * 4 + i] = (unsigned short)r[i];
}

}  // namespace mmdti
MMDTI_DEFINE_SALT_PULL(elementwise)
using namespace mmdti;

extern "C" const char* mmdti_last_error(void) { return g_err; }
extern "C" int mmdti_abi_version(void) { return 1; }

#define DROP_SETUP(name)                                                                  \
  MMDTI_REQUIRE(drop_p >= 0.f && drop_p < 1.f, name ": dropout p out of range");          \
  const uint32_t th = dropout_thresh(drop_p);                                             \
  const float sc = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f

extern "C" int mmdti_cast_f32_bf16(mmdti_stream_t stream, const float* x, void* y_bf16, long long n, float drop_p,
                                   unsigned long long seed, unsigned int site) {
  MMDTI_REQUIRE(x && y_bf16 && n > 0, "cast_f32_bf16: bad arguments");
  MMDTI_REQUIRE(aligned16(x) && (reinterpret_cast<uintptr_t>(y_bf16) & 7) == 0, "cast_f32_bf16: alignment");
  DROP_SETUP("cast_f32_bf16");
  hipLaunchKernelGGL(cast_f32_bf16_kernel<false>, dim3(grid_for(n / 4 + 1, 256)), dim3(256), 0, (hipStream_t)stream, x,
                     (bf16_t*)y_bf16, n / 4, n, th, sc, (uint64_t)seed, (uint32_t)site);
  MMDTI_LAUNCH_CHECK();
  return MMDTI_OK;
}

extern "C" int mmdti_cast_f32_f16(mmdti_stream_t stream, const float* x, void* y_f16, long long n, float drop_p,
                                  unsigned long long seed, unsigned int site) {
  MMDTI_REQUIRE(x && y_f16 && n > 0, "cast_f32_f16: bad arguments");
  MMDTI_REQUIRE(aligned16(x) && (reinterpret_cast<uintptr_t>(y_f16) & 7) == 0, "cast_f32_f16: alignment");
  DROP_SETUP("cast_f32_f16");
  hipLaunchKernelGGL(cast_f32_bf16_kernel<true>, dim3(grid_for(n / 4 + 1, 256)), dim3(256), 0, (hipStream_t)stream, x,
                     (bf16_t*)y_f16, n / 4, n, th, sc, (uint64_t)seed, (uint32_t)site);
  MMDTI_LAUNCH_CHECK();
  return MMDTI_OK;
}

extern "C" int mmdti_cast_f16_bf16(mmdti_stream_t stream, const void* x_f16, int rows, int cols, int ldx, void* y_bf16) {
  MMDTI_REQUIRE(x_f16 && y_bf16 && rows > 0 && cols > 0 && cols % 8 == 0 && ldx >= cols && ldx % 8 == 0, "cast_f16_bf16: bad arguments (cols %% 8, ldx %% 8)");
  MMDTI_REQUIRE(aligned16(x_f16) && aligned16(y_bf16), "cast_f16_bf16: 16-byte alignment required");
  const long long n8 = (long long)rows * (cols / 8);
  hipLaunchKernelGGL(cast_f16_bf16_kernel, dim3(grid_for(n8, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x_f16, cols / 8, ldx,
                     (bf16_t*)y_bf16, n8);
  MMDTI_LAUNCH_CHECK();
  return MMDTI_OK;
}

extern "C" int mmdti_cast_bf16_f32(mmdti_stream_t stream, const void* x_bf16, float* y, long long n) {
  MMDTI_REQUIRE(x_bf16 && y && n > 0, "cast_bf16_f32: bad arguments");
  hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x_bf16, y, n);
  MMDTI_LAUNCH_CHECK();
  return MMDTI_OK;
}

extern "C" int mmdti_dropout_f32(mmdti_stream_t stream, const float* x, float* y, long long n, float drop_p,
                                 unsigned long long seed, unsigned int site) {
  MMDTI_REQUIRE(x && y && n > 0, "dropout_f32: bad arguments");
  DROP_SETUP("dropout_f32");
  hipLaunchKernelGGL(dropout_f32_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, x, y, n, th, sc,
                     (uint64_t)seed, (uint32_t)site);
  MMDTI_LAUNCH_CHECK();
  return MMDTI_OK;
}

extern "C" int mmdti_axpy_f32(mmdti_stream_t stream, const float* x, float* y, long long n, float a) {
  MMDTI_REQUIRE(x && y && n > 0, "axpy_f32: bad arguments");
  hipLaunchKernelGGL(axpy_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, x, y, n, a);
  MMDTI_LAUNCH_CHECK();
  return MMDTI_OK;
}

extern "C" int mmdti_gelu_fwd_bf16(mmdti_stream_t stream, const void* u_bf16, void* y_bf16, long long n) {
  MMDTI_REQUIRE(u_bf16 && y_bf16 && n > 0, "gelu_fwd_bf16: bad arguments");
  hipLaunchKernelGGL(gelu_bf16_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)u_bf16, (bf16_t*)y_bf16, n);
  MMDTI_LAUNCH_CHECK();
  return MMDTI_OK;
}

extern "C" int mmdti_colsum_bf16(mmdti_stream_t stream, const void* x_bf16, int rows, int cols, int ld, float* out) {
  MMDTI_REQUIRE(x_bf16 && out && rows > 0 && cols > 0 && ld >= cols, "colsum_bf16: bad arguments");
  MMDTI_REQUIRE(ld % 8 == 0 && aligned16(x_bf16), "colsum_bf16: ld%%8 and 16-byte alignment required");
  // 8 rows per thread (two rounds of four 16-byte loads): enough workgroups to fill 256 CUs several times over --
  // the reduction is latency-bound at one workgroup per CU
  int gy = cdiv(rows, 8 * 8);
  if (gy > 1024) gy = 1024;
  hipLaunchKernelGGL(colsum_bf16_kernel, dim3(cdiv(cols, 256), gy), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x_bf16, rows, cols, ld, out);
  MMDTI_LAUNCH_CHECK();
  return MMDTI_OK;
}

extern "C" int mmdti_embedding_fwd(mmdti_stream_t stream, const long long* ids, const float* table, long long n,
                                   int D, int vocab, float* out, int accumulate) {
  MMDTI_REQUIRE(ids && table && out && n > 0 && D > 0 && D % 4 == 0 && vocab > 0, "embedding_fwd: bad arguments");
  MMDTI_REQUIRE(aligned16(table) && aligned16(out), "embedding_fwd: alignment");
  hipLaunchKernelGGL(embedding_fwd_kernel, dim3(grid_for(n * (D / 4), 256)), dim3(256), 0, (hipStream_t)stream, ids,
                     table, n, D / 4, vocab, out, accumulate);
  MMDTI_LAUNCH_CHECK();
  return MMDTI_OK;
}

extern "C" int mmdti_embedding_bwd(mmdti_stream_t stream, const long long* ids, const float* dout, long long n,
                                   int D, int vocab, long long padding_idx, float* dtable) {
  MMDTI_REQUIRE(ids && dout && dtable && n > 0 && D > 0 && vocab > 0, "embedding_bwd: bad arguments");
  hipLaunchKernelGGL(embedding_bwd_kernel, dim3(grid_for(n * D, 256)), dim3(256), 0, (hipStream_t)stream, ids, dout,
                     n, D, vocab, padding_idx, dtable);
  MMDTI_LAUNCH_CHECK();
  return MMDTI_OK;
}

extern "C" int mmdti_embedding_fwd3(mmdti_stream_t stream, const long long* ids_a, const float* table_a, int vocab_a, const long long* ids_b,
                                    const float* table_b, int vocab_b, const long long* ids_c, const float* table_c, int vocab_c, long long n,
                                    int D, float* out) {
  MMDTI_REQUIRE(ids_a && table_a && ids_b && table_b && table_c && out && n > 0 && D > 0 && D % 4 == 0 && vocab_a > 0 && vocab_b > 0 && vocab_c > 0,
                "embedding_fwd3: bad arguments");
  MMDTI_REQUIRE(aligned16(table_a) && aligned16(table_b) && aligned16(table_c) && aligned16(out), "embedding_fwd3: alignment");
  hipLaunchKernelGGL(embedding_fwd3_kernel, dim3(grid_for(n * (D / 4), 256)), dim3(256), 0, (hipStream_t)stream, ids_a, table_a, vocab_a, ids_b,
                     table_b, vocab_b, ids_c, table_c, vocab_c, n, D / 4, out);
  MMDTI_LAUNCH_CHECK();
  return MMDTI_OK;
}

extern "C" int mmdti_onehot_bf16(mmdti_stream_t stream, const long long* ids, long long n, int vocab, int ld,
                                 long long padding_idx, void* out_bf16) {
  MMDTI_REQUIRE(ids && out_bf16 && n > 0 && vocab > 0 && ld >= vocab && ld % 8 == 0, "onehot_bf16: bad arguments");
  MMDTI_REQUIRE(aligned16(out_bf16), "onehot_bf16: output must be 16-byte aligned");
  hipLaunchKernelGGL(onehot_bf16_kernel, dim3(grid_for(n * (ld / 8), 256)), dim3(256), 0, (hipStream_t)stream, ids, n,
                     vocab, ld / 8, padding_idx, (bf16_t*)out_bf16);
  MMDTI_LAUNCH_CHECK();
  return MMDTI_OK;
}

extern "C" int mmdti_roberta_position_ids(mmdti_stream_t stream, const long long* ids, int B, int L,
                                          long long pad_idx, long long* out) {
  MMDTI_REQUIRE(ids && out && B > 0 && L > 0, "roberta_position_ids: bad arguments");
  hipLaunchKernelGGL(roberta_posids_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, ids, L, pad_idx, out);
  MMDTI_LAUNCH_CHECK();
  return MMDTI_OK;
}

extern "C" int mmdti_softmax_fwd(mmdti_stream_t stream, const float* s, const float* key_add, void* p_bf16,
                                 void* pd_bf16, int B, int heads, int Lq, int Lk, int ld, float drop_p,
                                 unsigned long long seed, unsigned int site) {
  MMDTI_REQUIRE(s && p_bf16 && pd_bf16 && B > 0 && heads > 0 && Lq > 0 && Lk > 0, "softmax_fwd: bad arguments");
  MMDTI_REQUIRE(ld >= Lk && ld <= 1024 && ld % 8 == 0, "softmax_fwd: need Lk <= ld <= 1024, ld %% 8 == 0");
  MMDTI_REQUIRE(aligned16(s) && aligned16(p_bf16) && aligned16(pd_bf16), "softmax_fwd: 16-byte alignment required");
  DROP_SETUP("softmax_fwd");
  MMDTI_REQUIRE(drop_p == 0.f || pd_bf16 != p_bf16, "softmax_fwd: dropout needs a separate pd buffer");
  const long long rows = (long long)B * heads * Lq;
  dim3 grid(cdiv(rows, 4)), block(256);
#define SM_F(NV)                                                                                                   \
  hipLaunchKernelGGL((softmax_fwd_kernel<NV>), grid, block, 0, (hipStream_t)stream, s, key_add, (bf16_t*)p_bf16,  \
                     (bf16_t*)pd_bf16, rows, heads, Lq, Lk, ld, th, sc, (uint64_t)seed, (uint32_t)site)
  const int nv = (ld + 255) / 256;
  if (nv <= 1) SM_F(1); else if (nv <= 2) SM_F(2); else SM_F(4);
#undef SM_F
  MMDTI_LAUNCH_CHECK();
  return MMDTI_OK;
}

extern "C" int mmdti_softmax_bwd(mmdti_stream_t stream, const void* p_bf16, const float* dp, void* ds_bf16, int B,
                                 int heads, int Lq, int Lk, int ld, float scale, float drop_p,
                                 unsigned long long seed, unsigned int site) {
  MMDTI_REQUIRE(p_bf16 && dp && ds_bf16 && B > 0 && heads > 0 && Lq > 0 && Lk > 0, "softmax_bwd: bad arguments");
  MMDTI_REQUIRE(ld >= Lk && ld <= 1024 && ld % 8 == 0, "softmax_bwd: need Lk <= ld <= 1024, ld %% 8 == 0");
  MMDTI_REQUIRE(aligned16(dp) && aligned16(p_bf16) && aligned16(ds_bf16), "softmax_bwd: 16-byte alignment required");
  DROP_SETUP("softmax_bwd");
  const long long rows = (long long)B * heads * Lq;
  dim3 grid(cdiv(rows, 4)), block(256);
#define SM_B(NV)                                                                                              \
  hipLaunchKernelGGL((softmax_bwd_kernel<NV>), grid, block, 0, (hipStream_t)stream, (const bf16_t*)p_bf16, dp, \
                     (bf16_t*)ds_bf16, rows, Lk, ld, scale, th, sc, (uint64_t)seed, (uint32_t)site)
  const int nv = (ld + 255) / 256;
  if (nv <= 1) SM_B(1); else if (nv <= 2) SM_B(2); else SM_B(4);
#undef SM_B
  MMDTI_LAUNCH_CHECK();
  return MMDTI_OK;
}

extern "C" int mmdti_masked_pool_fwd(mmdti_stream_t stream, const float* a, const float* t,
                                     const unsigned char* mask_a, const unsigned char* mask_t, int B, int Na, int Nt,
                                     int D, float* pooled) {
  MMDTI_REQUIRE(a && t && mask_a && mask_t && pooled && B > 0 && Na > 0 && Nt > 0 && D > 0, "masked_pool_fwd: bad arguments");
  MMDTI_REQUIRE(D % 4 == 0 && aligned16(a) && aligned16(t) && aligned16(pooled), "masked_pool_fwd: D%%4 and 16-byte alignment required");
  hipLaunchKernelGGL(masked_pool_fwd_kernel, dim3(B, cdiv(D, 64)), dim3(256), 0, (hipStream_t)stream, a, t, mask_a, mask_t, Na, Nt, D, pooled);
  MMDTI_LAUNCH_CHECK();
  return MMDTI_OK;
}

extern "C" int mmdti_masked_pool_bwd(mmdti_stream_t stream, const float* dpooled, const unsigned char* mask_a,
                                     const unsigned char* mask_t, int B, int Na, int Nt, int D, float* da, float* dt) {
  MMDTI_REQUIRE(dpooled && mask_a && mask_t && da && dt && B > 0 && Na > 0 && Nt > 0 && D > 0, "masked_pool_bwd: bad arguments");
  MMDTI_REQUIRE(D % 4 == 0 && aligned16(dpooled) && aligned16(da) && aligned16(dt), "masked_pool_bwd: D%%4 and 16-byte alignment required");
  hipLaunchKernelGGL(masked_pool_bwd_kernel, dim3(B, 8), dim3(256), 0, (hipStream_t)stream, dpooled, mask_a, mask_t, Na, Nt, D, da, dt);
  MMDTI_LAUNCH_CHECK();
  return MMDTI_OK;
}

extern "C" int mmdti_masked_pool_packed_fwd(mmdti_stream_t stream, const float* a, const float* t, const int* a_off, const int* a_cnt,
                                            const int* t_off, const int* t_cnt, int B, int D, float* pooled) {
  MMDTI_REQUIRE(a && t && a_off && a_cnt && t_off && t_cnt && pooled && B > 0 && D > 0, "masked_pool_packed_fwd: bad arguments");
  MMDTI_REQUIRE(D % 4 == 0 && aligned16(a) && aligned16(t) && aligned16(pooled), "masked_pool_packed_fwd: D%%4 and 16-byte alignment required");
  hipLaunchKernelGGL(masked_pool_packed_fwd_kernel, dim3(B, cdiv(D, 64)), dim3(256), 0, (hipStream_t)stream, a, t, a_off, a_cnt, t_off, t_cnt, D, pooled);
  MMDTI_LAUNCH_CHECK();
  return MMDTI_OK;
}

extern "C" int mmdti_masked_pool_packed_bwd(mmdti_stream_t stream, const float* dpooled, const int* a_off, const int* a_cnt, const int* a_seq,
                                            int rows_a, const int* t_off, const int* t_cnt, const int* t_seq, int rows_t, int D, float* da,
                                            float* dt) {
  MMDTI_REQUIRE(dpooled && a_off && a_cnt && a_seq && t_off && t_cnt && t_seq && da && dt && rows_a > 0 && rows_t > 0 && D > 0,
                "masked_pool_packed_bwd: bad arguments");
  MMDTI_REQUIRE(D % 4 == 0 && aligned16(dpooled) && aligned16(da) && aligned16(dt), "masked_pool_packed_bwd: D%%4 and 16-byte alignment required");
  const long long n = ((long long)rows_a + rows_t) * (D / 4);
  hipLaunchKernelGGL(masked_pool_packed_bwd_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, dpooled, a_off, a_cnt, a_seq,
                     rows_a, t_off, t_cnt, t_seq, rows_t, D, da, dt);
  MMDTI_LAUNCH_CHECK();
  return MMDTI_OK;
}

extern "C" int mmdti_sumsq_f32(mmdti_stream_t stream, const float* g, long long n, float* out, float* ws, int ws_floats) {
  MMDTI_REQUIRE(g && out && n > 0, "sumsq_f32: bad arguments");
  if (ws && ws_floats >= 1 && aligned16(g)) {
    const int wgs = (int)min((long long)min(ws_floats, SUMSQ_WGS), (n / 4 + 255) / 256 + 1);
    hipLaunchKernelGGL(sumsq_part_kernel, dim3(wgs), dim3(256), 0, (hipStream_t)stream, g, n, ws);
    hipLaunchKernelGGL(sumsq_fold_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, ws, wgs, out);
  } else {
    hipLaunchKernelGGL(sumsq_kernel, dim3(grid_for(n, 256 * 8)), dim3(256), 0, (hipStream_t)stream, g, n, out);
  }
  MMDTI_LAUNCH_CHECK();
  return MMDTI_OK;
}

extern "C" int mmdti_adam_step(mmdti_stream_t stream, float* p, const float* g, float* m, float* v, void* p_bf16,
                               long long n, float lr, float beta1, float beta2, float eps, float weight_decay,
                               int step, const float* grad_scale_dev, const float* step_state_dev, void* p_f16) {
  MMDTI_REQUIRE(p && g && m && v && n > 0 && (step >= 1 || step_state_dev), "adam_step: bad arguments");
  const float bc1 = 1.f - powf(beta1, (float)step);
  const float bc2s = sqrtf(1.f - powf(beta2, (float)step));
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n, 256 * 4)), dim3(256), 0, (hipStream_t)stream, p, g, m, v,
                     (bf16_t*)p_bf16, (bf16_t*)p_f16, n, lr, beta1, beta2, eps, weight_decay, bc1, bc2s, grad_scale_dev, step_state_dev);
  MMDTI_LAUNCH_CHECK();
  return MMDTI_OK;
}

extern "C" int mmdti_step_state_advance(mmdti_stream_t stream, float* state, unsigned long long* salt, float base_lr, int warmup_steps,
                                        int total_steps, float beta1, float beta2) {
  MMDTI_REQUIRE(state && salt, "step_state_advance: null pointer");
  MMDTI_REQUIRE(total_steps > 0 && warmup_steps >= 0, "step_state_advance: bad schedule");
  hipLaunchKernelGGL(step_state_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, state, salt, base_lr, warmup_steps, total_steps, beta1, beta2);
  MMDTI_LAUNCH_CHECK();
  return MMDTI_OK;
}

extern "C" int mmdti_seed_salt_pull(mmdti_stream_t stream, const unsigned long long* salt) {
  MMDTI_REQUIRE(salt != nullptr, "seed_salt_pull: null pointer");
  hipStream_t s = (hipStream_t)stream;
  salt_pull_gemm(s, salt);
  salt_pull_layernorm(s, salt);
  salt_pull_pair_attn(s, salt);
  salt_pull_pair_attn_bwd(s, salt);
  salt_pull_pair_attn_bwd_g16(s, salt);
  salt_pull_attn(s, salt);
  salt_pull_elementwise(s, salt);
  MMDTI_LAUNCH_CHECK();
  return MMDTI_OK;
}

extern "C" int mmdti_probe_tr_read(mmdti_stream_t stream, int row_stride_elems, unsigned short* out) {
  MMDTI_REQUIRE(out && row_stride_elems >= 16 && row_stride_elems <= 64 && row_stride_elems % 4 == 0, "probe_tr_read: bad stride");
  hipLaunchKernelGGL(probe_tr_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, row_stride_elems, out);
  MMDTI_LAUNCH_CHECK();
  return MMDTI_OK;
}
