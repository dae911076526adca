// pair_attn_bwd.hip -- backward of the pair-bias attention (see pair_attn.hip for the forward and the layouts).
#include "pair_attn_bwd_mfma.h"

namespace mmdti {

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
}  // namespace mmdti
MMDTI_DEFINE_SALT_PULL(pair_attn_bwd)
using namespace mmdti;

extern "C" int mmdti_pair_attn_bwd(mmdti_stream_t stream, const void* qkv_bf16, const void* s, const void* do_bf16,
                                   void* g, void* dqkv_bf16, int B, int N, int H, int ld, float scale,
                                   int g_in_zero, float drop_p, unsigned long long seed, unsigned int site, int layout,
                                   const int* key_tiles, const int* row_off, int qkv_f16) {
  // bit 0: tiled planes; bit 1: compact planes (s is fp16; tiled only); bit 2: g is bf16 (with bit 1 only)
  const int tiled = layout & 1, compact = (layout >> 1) & 1, g16 = (layout >> 2) & 1;
  if (int e = check_common("pair_attn_bwd", B, N, H, ld)) return e;
  MMDTI_REQUIRE((layout & ~7) == 0 && (!compact || tiled) && (!g16 || compact),
                "pair_attn_bwd: layout must be 0 (row-major fp32), 1 (tiled fp32), 3 (tiled, fp16 logits) or 7 (tiled, fp16 logits, bf16 gradients)");
  MMDTI_REQUIRE(!key_tiles || compact, "pair_attn_bwd: key_tiles (ragged batches) needs the compact planes (layout 3 or 7: tiled, fp16 logits)");
  MMDTI_REQUIRE(!row_off || key_tiles, "pair_attn_bwd: packed token rows (row_off) need key_tiles");
  MMDTI_REQUIRE(!tiled || (ld % 4 == 0 && N <= 16 * PA_MAX_NT), "pair_attn_bwd: the tiled pair layout needs ld %% 4 == 0 and N <= 272");
  MMDTI_REQUIRE(qkv_bf16 && s && do_bf16 && g && dqkv_bf16, "pair_attn_bwd: null pointer");
  MMDTI_REQUIRE(aligned16(qkv_bf16) && aligned16(do_bf16) && aligned16(dqkv_bf16), "pair_attn_bwd: alignment");
  MMDTI_REQUIRE(!tiled || (aligned16(s) && aligned16(g)), "pair_attn_bwd: tiled pair tensors must be 16-byte aligned");
  const uint32_t th = dropout_thresh(drop_p);     // the per-element kernel: Philox words against p * 2^32
  const uint32_t th8 = dropout_thresh8(drop_p);   // the MFMA kernel: the shared byte generator of common.h (as its forward)
  const float sc = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  dim3 grid(B * H), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (ld % 4 == 0 && aligned16(s) && aligned16(g) && N <= 16 * PA_MAX_NT) {
    const int nqb = (N + 15) / 16;
    // (16 and 17 tiles: three waves and a register cap for two workgroups per CU -- with four waves the per-wave dK / dV accumulators
    //  leave LDS for one workgroup, and 251 + 12 registers for one wave per SIMD: N = 250 / 258 took 4.2 / 3.9 ms against 2.15 at N = 240)
    dim3 blk((nqb % 3 == 0 || nqb >= 16 || nqb == 5) ? 192 : (nqb < 4 ? 64 * nqb : 256));   // (5 blocks: 2,2,1 on three waves beats 2,1,1,1 on four)
#define PA_MB(NT, TL, FL, NWV, RG, ST, GT)                                                                                     \
  hipLaunchKernelGGL((pair_attn_bwd_mfma_kernel<NT, TL, FL, NWV, RG, ST, GT>), grid, blk, 0, st, (const bf16_t*)qkv_bf16,      \
                     (const ST*)s, (const bf16_t*)do_bf16, (const GT*)g, (GT*)g, (bf16_t*)dqkv_bf16, N, H, ld, scale,           \
                     g_in_zero, th8, sc, (uint64_t)seed, (uint32_t)site, key_tiles, row_off, qkv_f16)
#define PA_MBW(NT, TL, FL, RG, ST, GT)                                                      \
  do {                                                                                      \
    if (blk.x == 192) PA_MB(NT, TL, FL, 3, RG, ST, GT); else PA_MB(NT, TL, FL, 4, RG, ST, GT); \
  } while (0)
#define PA_MBT(NT)                                                                          \
  do {                                                                                      \
    if (!tiled) PA_MBW(NT, false, false, false, float, float);                              \
    else { if (nqb == NT) PA_MBW(NT, true, true, false, float, float); else PA_MBW(NT, true, false, false, float, float); } \
  } while (0)
    // (compact planes -- the hot path -- have one instantiation per tile count, for fp32 and for bf16 gradients: pair_attn_bwd_mfma.h)
    if (compact) {
      if (g16) pa_bwd_compact_launch_g16(nqb, grid, blk, st, qkv_bf16, s, do_bf16, g, dqkv_bf16, N, H, ld, scale, g_in_zero, th8, sc, seed, site, key_tiles, row_off, qkv_f16);
      else pa_bwd_compact_launch<float>(nqb, grid, blk, st, qkv_bf16, s, do_bf16, g, dqkv_bf16, N, H, ld, scale, g_in_zero, th8, sc, seed, site, key_tiles, row_off, qkv_f16);
    } else if (nqb <= 5) PA_MBT(5); else if (nqb <= 9) PA_MBT(9); else if (nqb <= 13) PA_MBT(13); else PA_MBT(17);
#undef PA_MBW
#undef PA_MBT
#undef PA_MB
    MMDTI_LAUNCH_CHECK();
    return MMDTI_OK;
  }
  MMDTI_REQUIRE(!qkv_f16, "pair_attn_bwd: fp16 q | k | v are read by the MFMA kernels only (ld %% 4 == 0, N <= 272, 16-byte aligned pair tensors)");
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
