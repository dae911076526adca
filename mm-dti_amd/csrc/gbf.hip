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
                                                               int N, int H, int ld) {
  extern __shared__ float tile[];  // [H][NP], NP odd
  const int bi = blockIdx.x, b = bi / N, i = bi - b * N;
  const int NP = N | 1;
  for (int t = threadIdx.x; t < H * N; t += 256) {
    const int h = t / N, j = t - h * N;
    tile[h * NP + j] = g[(((long long)b * H + h) * N + i) * ld + j];
  }
  __syncthreads();
  bf16_t* dst = out + (long long)bi * N * H;
  for (int t = threadIdx.x; t < N * H; t += 256) {
    const int j = t / H, h = t - j * H;
    dst[t] = f2bf(tile[h * NP + j]);
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
                                      int ld) {
  MMDTI_REQUIRE(g && out_bf16 && B > 0 && N > 0 && H > 0 && ld >= N, "pair_permute_bwd: bad arguments");
  const size_t smem = (size_t)H * (N | 1) * sizeof(float);
  MMDTI_REQUIRE(smem <= 64 * 1024, "pair_permute_bwd: N*H too large for the LDS tile (N=%d,H=%d)", N, H);
  hipLaunchKernelGGL(pair_permute_bwd_kernel, dim3(B * N), dim3(256), smem, (hipStream_t)stream, g, (bf16_t*)out_bf16,
                     N, H, ld);
  MMDTI_LAUNCH_CHECK();
  return MMDTI_OK;
}
