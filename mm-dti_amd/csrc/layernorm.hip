// LayerNorm forward / backward, one wave per row, rows of D floats (D % 4 == 0, D <= 2048).
// Replaces unicore.modules.LayerNorm (models/transformers.py:69,71,76 + 2 per encoder layer), BertLayerNorm
// (models/mm_module.py:320-333) and HF nn.LayerNorm in RobertaModel.  HBM-bound: fwd reads 4 B and writes 2-6 B
// per element; float4 loads, statistics in fp32 registers (two-pass, biased variance, eps inside the sqrt).
#include "common.h"

namespace mmdti {

// up to 8 float4 per lane -> D <= 64*4*8 = 2048

template <int NV>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float eps, int rows, int D,
                                                     float* __restrict__ y32, bf16_t* __restrict__ y16,
                                                     float* __restrict__ mean_o, float* __restrict__ rstd_o,
                                                     const unsigned char* __restrict__ row_zero, uint32_t thresh,
                                                     float dscale, uint64_t seed, uint32_t site, int y16_f16) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nvec = D >> 2;
  const float4* xr = reinterpret_cast<const float4*>(x + (long long)row * D);
  float4 v[NV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    int c = lane + i * 64;
    v[i] = (c < nvec) ? xr[c] : make_float4(0.f, 0.f, 0.f, 0.f);
    s += v[i].x + v[i].y + v[i].z + v[i].w;
  }
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    int c = lane + i * 64;
    if (c < nvec) {
      float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
      q += a * a + b * b + cc * cc + d * d;
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
  if (lane == 0) {
    if (mean_o) mean_o[row] = mean;
    if (rstd_o) rstd_o[row] = rstd;
  }
  const bool zero = row_zero && row_zero[row];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    int c = lane + i * 64;
    if (c >= nvec) continue;
    float4 g = reinterpret_cast<const float4*>(gamma)[c], b = reinterpret_cast<const float4*>(beta)[c];
    float o[4] = {(v[i].x - mean) * rstd * g.x + b.x, (v[i].y - mean) * rstd * g.y + b.y,
                  (v[i].z - mean) * rstd * g.z + b.z, (v[i].w - mean) * rstd * g.w + b.w};
    if (thresh) {
      Rand4 r = philox4(seed, site, ((uint64_t)row * D + (uint64_t)c * 4) >> 2);
      uint32_t rw[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = rw[e] >= thresh ? o[e] * dscale : 0.f;
    }
    if (zero) o[0] = o[1] = o[2] = o[3] = 0.f;
    if (y32) reinterpret_cast<float4*>(y32 + (long long)row * D)[c] = make_float4(o[0], o[1], o[2], o[3]);
    if (y16) {
      uint2 pk;
      if (y16_f16) {   // fp16 forward-operand mode
        pk.x = f2h_sat2(o[0], o[1]);
        pk.y = f2h_sat2(o[2], o[3]);
      } else {
        pk.x = (uint32_t)f2bf(o[0]) | ((uint32_t)f2bf(o[1]) << 16);
        pk.y = (uint32_t)f2bf(o[2]) | ((uint32_t)f2bf(o[3]) << 16);
      }
      reinterpret_cast<uint2*>(y16 + (long long)row * D)[c] = pk;
    }
  }
}

// Backward.  Each block handles ROWS_PER_BLOCK rows (4 waves x 8 rows); dgamma/dbeta partials are kept per lane in
// registers over the wave's rows, combined across the 4 waves in LDS and added to global with one atomic per column.
constexpr int LN_BWD_MIN_ROWS_PER_WAVE = 4;

template <int NV, bool DY_BF16>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const void* __restrict__ dy_, const float* __restrict__ dy_add,
                                                     const float* __restrict__ x,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean_i,
                                                     const float* __restrict__ rstd_i, int rows, int D,
                                                     const float* __restrict__ dres, float* __restrict__ dx,
                                                     float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                     const unsigned char* __restrict__ row_zero, uint32_t thresh,
                                                     float dscale, uint64_t seed, uint32_t site,
                                                     bf16_t* __restrict__ dx_bf16, uint32_t thresh2, float dscale2,
                                                     uint32_t site2, float* __restrict__ dx_colsum, int rpw) {
  extern __shared__ __attribute__((aligned(16))) float red[];  // [3][4][D]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nvec = D >> 2;
  float4 ag[NV], ab[NV], ac[NV];   // dgamma, dbeta, column sums of the bf16 copy (the next Linear's bias gradient)
  float4 gm[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    ag[i] = ab[i] = ac[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int c = lane + i * 64;
    gm[i] = c < nvec ? reinterpret_cast<const float4*>(gamma)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const int row_base = (blockIdx.x * 4 + wave) * rpw;
  const int nrow = min(rpw, rows - row_base);
  // Every byte of a row is requested up front (dy, dy_add, x AND the residual gradient), as raw words, and the next
  // row's requests go out before this row is reduced: two rows per wave in flight instead of a row's first half.
  struct RowIn {
    float4 d32[NV], da[NV], xv[NV], rs[NV];
    uint2 d16[NV];
    float mean, rstd;
    bool zero;
  };
  auto fetch = [&](int row, RowIn& R) {
    R.mean = mean_i[row];
    R.rstd = rstd_i[row];
    R.zero = row_zero && row_zero[row];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane + i * 64;
      R.d32[i] = R.da[i] = R.xv[i] = R.rs[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      R.d16[i] = make_uint2(0u, 0u);
      if (c >= nvec) continue;
      if (DY_BF16) R.d16[i] = reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(dy_) + (long long)row * D)[c];
      else R.d32[i] = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(dy_) + (long long)row * D)[c];
      if (dy_add) R.da[i] = reinterpret_cast<const float4*>(dy_add + (long long)row * D)[c];
      R.xv[i] = reinterpret_cast<const float4*>(x + (long long)row * D)[c];
      if (dres) R.rs[i] = reinterpret_cast<const float4*>(dres + (long long)row * D)[c];
    }
  };
  RowIn cur, nxt;
  if (nrow > 0) fetch(row_base, cur);
  for (int rr = 0; rr < nrow; ++rr) {
    const int row = row_base + rr;
    if (rr + 1 < nrow) fetch(row + 1, nxt);
    const float mean = cur.mean, rstd = cur.rstd;
    float4 xh[NV], dg[NV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int c = lane + i * 64;
      xh[i] = dg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c >= nvec) continue;
      float d[4];
      if (DY_BF16) {
        const uint2 pk = cur.d16[i];
        d[0] = bf2f((bf16_t)(pk.x & 0xffff)); d[1] = bf2f((bf16_t)(pk.x >> 16));
        d[2] = bf2f((bf16_t)(pk.y & 0xffff)); d[3] = bf2f((bf16_t)(pk.y >> 16));
      } else {
        d[0] = cur.d32[i].x; d[1] = cur.d32[i].y; d[2] = cur.d32[i].z; d[3] = cur.d32[i].w;
      }
      if (dy_add) {
        d[0] += cur.da[i].x; d[1] += cur.da[i].y; d[2] += cur.da[i].z; d[3] += cur.da[i].w;
      }
      if (thresh) {
        Rand4 r = philox4(seed, site, ((uint64_t)row * D + (uint64_t)c * 4) >> 2);
        uint32_t rw[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) d[e] = rw[e] >= thresh ? d[e] * dscale : 0.f;
      }
      if (cur.zero) d[0] = d[1] = d[2] = d[3] = 0.f;
      const float4 xv = cur.xv[i], g = gm[i];
      xh[i] = make_float4((xv.x - mean) * rstd, (xv.y - mean) * rstd, (xv.z - mean) * rstd, (xv.w - mean) * rstd);
      ag[i].x += d[0] * xh[i].x; ag[i].y += d[1] * xh[i].y; ag[i].z += d[2] * xh[i].z; ag[i].w += d[3] * xh[i].w;
      ab[i].x += d[0]; ab[i].y += d[1]; ab[i].z += d[2]; ab[i].w += d[3];
      dg[i] = make_float4(d[0] * g.x, d[1] * g.y, d[2] * g.z, d[3] * g.w);
      s1 += dg[i].x + dg[i].y + dg[i].z + dg[i].w;
      s2 += dg[i].x * xh[i].x + dg[i].y * xh[i].y + dg[i].z * xh[i].z + dg[i].w * xh[i].w;
    }
    const float m1 = wave_sum(s1) / (float)D, m2 = wave_sum(s2) / (float)D;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int c = lane + i * 64;
      if (c >= nvec) continue;
      float4 o = make_float4(rstd * (dg[i].x - m1 - xh[i].x * m2), rstd * (dg[i].y - m1 - xh[i].y * m2),
                             rstd * (dg[i].z - m1 - xh[i].z * m2), rstd * (dg[i].w - m1 - xh[i].w * m2));
      if (dres) {
        const float4 r = cur.rs[i];
        o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
      }
      reinterpret_cast<float4*>(dx + (long long)row * D)[c] = o;
      if (dx_bf16) {  // the next GEMM's operand: dropout-backward of ANOTHER site applied to dx, rounded to bf16
        if (thresh2) {
          const Rand4 r = philox4(seed, site2, ((uint64_t)row * D + (uint64_t)c * 4) >> 2);
          o.x = r.x >= thresh2 ? o.x * dscale2 : 0.f; o.y = r.y >= thresh2 ? o.y * dscale2 : 0.f;
          o.z = r.z >= thresh2 ? o.z * dscale2 : 0.f; o.w = r.w >= thresh2 ? o.w * dscale2 : 0.f;
        }
        uint2 pk;
        pk.x = (uint32_t)f2bf(o.x) | ((uint32_t)f2bf(o.y) << 16);
        pk.y = (uint32_t)f2bf(o.z) | ((uint32_t)f2bf(o.w) << 16);
        reinterpret_cast<uint2*>(dx_bf16 + (long long)row * D)[c] = pk;
        if (dx_colsum) {   // sums the ROUNDED values: exactly what a column-sum pass over dx_bf16 would add up
          ac[i].x += __uint_as_float(pk.x << 16); ac[i].y += __uint_as_float(pk.x & 0xffff0000u);
          ac[i].z += __uint_as_float(pk.y << 16); ac[i].w += __uint_as_float(pk.y & 0xffff0000u);
        }
      }
    }
    cur = nxt;
  }
  // cross-wave reduction of dgamma/dbeta partials
  float* rg = red + wave * D;
  float* rb = red + 4 * D + wave * D;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    int c = lane + i * 64;
    if (c < nvec) {
      reinterpret_cast<float4*>(rg)[c] = ag[i];
      reinterpret_cast<float4*>(rb)[c] = ab[i];
    }
  }
  if (dx_colsum) {
    float* rc = red + 8 * D + wave * D;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int c = lane + i * 64;
      if (c < nvec) reinterpret_cast<float4*>(rc)[c] = ac[i];
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < D; c += 256) {
    float g = red[c] + red[D + c] + red[2 * D + c] + red[3 * D + c];
    float b = red[4 * D + c] + red[5 * D + c] + red[6 * D + c] + red[7 * D + c];
    if (dgamma) atomicAdd(dgamma + c, g);
    if (dbeta) atomicAdd(dbeta + c, b);
    if (dx_colsum) atomicAdd(dx_colsum + c, red[8 * D + c] + red[9 * D + c] + red[10 * D + c] + red[11 * D + c]);
  }
}

}  // namespace mmdti
MMDTI_DEFINE_SALT_PULL(layernorm)
using namespace mmdti;

static int ln_nv(int D) { return (D / 4 + 63) / 64; }

extern "C" int mmdti_layernorm_fwd(mmdti_stream_t stream, const float* x, const float* gamma, const float* beta,
                                   float eps, int rows, int D, float* y_f32, void* y_bf16, float* mean, float* rstd,
                                   const unsigned char* row_zero, float drop_p, unsigned long long seed,
                                   unsigned int site, int y16_f16) {
  MMDTI_REQUIRE(rows > 0 && D > 0 && D % 4 == 0 && D <= 2048, "layernorm_fwd: need rows>0, D%%4==0, D<=2048 (D=%d)", D);
  MMDTI_REQUIRE(x && gamma && beta && (y_f32 || y_bf16), "layernorm_fwd: null pointer");
  MMDTI_REQUIRE(aligned16(x) && aligned16(gamma) && aligned16(beta), "layernorm_fwd: 16-byte alignment required");
  MMDTI_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "layernorm_fwd: dropout p out of range");
  const uint32_t th = dropout_thresh(drop_p);
  const float sc = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  dim3 grid(cdiv(rows, 4)), block(256);
  hipStream_t s = (hipStream_t)stream;
#define LN_F(NV)                                                                                                    \
  hipLaunchKernelGGL((ln_fwd_kernel<NV>), grid, block, 0, s, x, gamma, beta, eps, rows, D, y_f32, (bf16_t*)y_bf16, \
                     mean, rstd, row_zero, th, sc, (uint64_t)seed, (uint32_t)site, y16_f16)
  switch (ln_nv(D)) {
    case 1: LN_F(1); break;
    case 2: LN_F(2); break;
    case 3: case 4: LN_F(4); break;
    default: LN_F(8); break;
  }
#undef LN_F
  MMDTI_LAUNCH_CHECK();
  return MMDTI_OK;
}

extern "C" int mmdti_layernorm_bwd(mmdti_stream_t stream, const void* dy, int dy_dtype, const float* dy_add, const float* x,
                                   const float* gamma, const float* mean, const float* rstd, int rows, int D,
                                   const float* dres, float* dx, float* dgamma, float* dbeta,
                                   const unsigned char* row_zero, float drop_p, unsigned long long seed,
                                   unsigned int site, void* dx_bf16, float drop2_p, unsigned int site2, float* dx_colsum) {
  MMDTI_REQUIRE(rows > 0 && D > 0 && D % 4 == 0 && D <= 2048, "layernorm_bwd: need rows>0, D%%4==0, D<=2048 (D=%d)", D);
  MMDTI_REQUIRE(dy && x && gamma && mean && rstd && dx, "layernorm_bwd: null pointer");
  MMDTI_REQUIRE(dy_dtype == MMDTI_DT_F32 || dy_dtype == MMDTI_DT_BF16, "layernorm_bwd: bad dy dtype");
  MMDTI_REQUIRE(aligned16(x) && aligned16(gamma) && aligned16(dx) && aligned16(dy), "layernorm_bwd: alignment");
  MMDTI_REQUIRE(drop_p >= 0.f && drop_p < 1.f && drop2_p >= 0.f && drop2_p < 1.f, "layernorm_bwd: dropout p out of range");
  MMDTI_REQUIRE(!dx_bf16 || (reinterpret_cast<uintptr_t>(dx_bf16) & 7) == 0, "layernorm_bwd: dx_bf16 alignment");
  const uint32_t th = dropout_thresh(drop_p);
  const float sc = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  const uint32_t th2 = dropout_thresh(drop2_p);
  const float sc2 = drop2_p > 0.f ? 1.f / (1.f - drop2_p) : 1.f;
  // rows per wave: enough that the whole grid is resident at once (3 workgroups of 4 waves per CU at this kernel's
  // register count, 2 for the wide-row variants) -- no second, partly filled round, and fewer dgamma / dbeta atomics
  const int rpw = max(LN_BWD_MIN_ROWS_PER_WAVE, cdiv(rows, 4 * (ln_nv(D) <= 2 ? 3 : 2) * 256));
  dim3 grid(cdiv(rows, 4 * rpw)), block(256);
  MMDTI_REQUIRE(!dx_colsum || dx_bf16, "layernorm_bwd: dx_colsum sums the bf16 copy, which was not requested");
  const size_t smem = 12 * (size_t)D * sizeof(float);
  hipStream_t s = (hipStream_t)stream;
#define LN_B(NV, BF)                                                                                               \
  hipLaunchKernelGGL((ln_bwd_kernel<NV, BF>), grid, block, smem, s, dy, dy_add, x, gamma, mean, rstd, rows, D, dres, dx,  \
                     dgamma, dbeta, row_zero, th, sc, (uint64_t)seed, (uint32_t)site, (bf16_t*)dx_bf16, th2, sc2, (uint32_t)site2, dx_colsum, rpw)
  const bool bf = dy_dtype == MMDTI_DT_BF16;
  switch (ln_nv(D)) {
    case 1: if (bf) LN_B(1, true); else LN_B(1, false); break;
    case 2: if (bf) LN_B(2, true); else LN_B(2, false); break;
    case 3: case 4: if (bf) LN_B(4, true); else LN_B(4, false); break;
    default: if (bf) LN_B(8, true); else LN_B(8, false); break;
  }
#undef LN_B
  MMDTI_LAUNCH_CHECK();
  return MMDTI_OK;
}
