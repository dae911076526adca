// bf16 MFMA GEMM for gfx950 with fused epilogues.
//
//   C[M,N] = epilogue( alpha * sum_k A[m,k] * B[n,k] )
//
// Replaces every nn.Linear / bmm on the hot path (SURVEY.md 8a rows a3, a6, a7,
// a8, a10): unicore TransformerEncoderLayer in_proj/out_proj/fc1/fc2 called from
// models/transformers.py:137-139, gbf_proj (mm_model.py:554), HF RobertaModel
// linears (mm_model.py:562), InfoNCE projections (infonce.py:20-21,28-29) and the
// BertCrossEncoder linears (mm_module.py:470-587), forward and backward.
//
// Tiling: 128x128x64 per 256-thread workgroup (4 waves as 2x2, each wave a 64x64
// sub-tile = 4x4 v_mfma_f32_16x16x32_bf16 accumulators), LDS double-buffered,
// one barrier per K-step.  Operands are staged global -> VGPR -> LDS in the
// canonical [row][k] image (row stride 72 bf16 = 144 B, conflict-reducing pad);
// an operand stored k-major (transposed) is transposed on the LDS write.
// XCD-aware block remap keeps tiles that share an A row-panel on one XCD's L2.
#include "common.h"

namespace mmdti {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int LDT = BK + 8;  // LDS row stride in elements (144 B, 16-B aligned)

struct GemmArgs {
  const bf16_t* A;
  const bf16_t* B;
  void* C;
  int M, N, K;
  int lda, ldb, ldc;
  int batch_inner;  // z = outer * batch_inner + inner
  long long sAo, sAi, sBo, sBi, sCo, sCi;
  int splitk;
  float alpha, beta;
  const float* bias;      // [N] fp32 or null
  const float* residual;  // fp32 [M, ldr] or null (unbatched only)
  int ldr;
  int act;                // MMDTI_ACT_*
  const bf16_t* aux_in;   // gelu_bwd: pre-activation u, [M, ld_aux] (+ batch strides of C scaled? unbatched only)
  bf16_t* aux_out;        // gelu: store pre-activation
  int ld_aux;
  int c_dtype;            // MMDTI_DT_F32 / MMDTI_DT_BF16 / MMDTI_DT_F32_ATOMIC
  uint32_t drop_thresh;
  float drop_scale;
  uint64_t seed;
  uint32_t site;
};

// Load one 128x64 operand tile slice into registers: 4 x 16-B chunks per thread.
//  !TR: memory is [rows][k] (k contiguous): chunk = 8 consecutive k of one row.
//   TR: memory is [k][rows] (rows contiguous): chunk = 8 consecutive rows of one k.
template <bool TR>
__device__ __forceinline__ void load_tile(uint4 (&r)[4], const bf16_t* __restrict__ base, int ld, int row0, int k0,
                                          int rows, int K, int tid) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int c = tid + i * 256;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (!TR) {
      int row = row0 + (c >> 3), k = k0 + (c & 7) * 8;
      if (row < rows && k < K) v = *reinterpret_cast<const uint4*>(base + (long long)row * ld + k);
    } else {
      int k = k0 + (c >> 4), row = row0 + (c & 15) * 8;
      if (k < K && row < rows) v = *reinterpret_cast<const uint4*>(base + (long long)k * ld + row);
    }
    r[i] = v;
  }
}

template <bool TR>
__device__ __forceinline__ void store_tile(const uint4 (&r)[4], bf16_t* lds, int tid) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int c = tid + i * 256;
    if (!TR) {
      int row = c >> 3, kc = (c & 7) * 8;
      *reinterpret_cast<uint4*>(lds + row * LDT + kc) = r[i];
    } else {
      int k = c >> 4, row = (c & 15) * 8;
      const bf16_t* e = reinterpret_cast<const bf16_t*>(&r[i]);
#pragma unroll
      for (int j = 0; j < 8; ++j) lds[(row + j) * LDT + k] = e[j];
    }
  }
}

template <bool TA, bool TB>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(GemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* smem = reinterpret_cast<bf16_t*>(smem_raw);
  bf16_t* sA[2] = {smem, smem + 2 * BM * LDT};
  bf16_t* sB[2] = {smem + BM * LDT, smem + 2 * BM * LDT + BM * LDT};

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;

  // XCD-aware remap of the (n-tile, m-tile) index: consecutive hardware block ids go round-robin over the 8 XCDs,
  // so give each XCD a contiguous chunk of the tile list (bijective form, cdna guide T1).
  const int tiles_n = (a.N + BN - 1) / BN, tiles_m = (a.M + BM - 1) / BM;
  const int nwg = tiles_n * tiles_m;
  int orig = blockIdx.x;
  int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
  int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  const int tm = wg / tiles_n, tn = wg % tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  const int z = blockIdx.z;
  const int zb = z / a.splitk, ks = z - zb * a.splitk;
  const int zo = zb / a.batch_inner, zi = zb - zo * a.batch_inner;
  const bf16_t* A = a.A + zo * a.sAo + zi * a.sAi;
  const bf16_t* B = a.B + zo * a.sBo + zi * a.sBi;

  // K range of this split
  const int ktiles = (a.K + BK - 1) / BK;
  const int per = (ktiles + a.splitk - 1) / a.splitk;
  const int kt0 = ks * per, kt1 = min(ktiles, kt0 + per);

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  uint4 ra[4], rb[4];
  if (kt0 < kt1) {
    load_tile<TA>(ra, A, a.lda, m0, kt0 * BK, a.M, a.K, tid);
    load_tile<TB>(rb, B, a.ldb, n0, kt0 * BK, a.N, a.K, tid);
    store_tile<TA>(ra, sA[0], tid);
    store_tile<TB>(rb, sB[0], tid);
  }
  __syncthreads();

  int cur = 0;
  for (int kt = kt0; kt < kt1; ++kt) {
    const bool more = (kt + 1 < kt1);
    if (more) {
      load_tile<TA>(ra, A, a.lda, m0, (kt + 1) * BK, a.M, a.K, tid);
      load_tile<TB>(rb, B, a.ldb, n0, (kt + 1) * BK, a.N, a.K, tid);
    }
    const bf16_t* cA = sA[cur] + (wr * 64 + (lane & 15)) * LDT + (lane >> 4) * 8;
    const bf16_t* cB = sB[cur] + (wc * 64 + (lane & 15)) * LDT + (lane >> 4) * 8;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        fa[i] = *reinterpret_cast<const bf16x8*>(cA + i * 16 * LDT + kk * 32);
        fb[i] = *reinterpret_cast<const bf16x8*>(cB + i * 16 * LDT + kk * 32);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    if (more) {
      store_tile<TA>(ra, sA[cur ^ 1], tid);
      store_tile<TB>(rb, sB[cur ^ 1], tid);
    }
    __syncthreads();
    cur ^= 1;
  }
  if (kt0 >= kt1 && a.splitk > 1) return;  // empty split contributes nothing

  // ---- epilogue: C/D layout col = lane&15, row = (lane>>4)*4 + reg ----
  const long long coff = zo * a.sCo + zi * a.sCi;
  const bool lead = (ks == 0);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = n0 + wc * 64 + j * 16 + (lane & 15);
      if (col >= a.N) continue;
      const float bv = (a.bias && lead) ? a.bias[col] : 0.f;
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int row = m0 + wr * 64 + i * 16 + (lane >> 4) * 4 + rr;
        if (row >= a.M) continue;
        float v = acc[i][j][rr] * a.alpha + bv;
        if (a.act == MMDTI_ACT_GELU) {
          if (a.aux_out) a.aux_out[(long long)row * a.ld_aux + col] = f2bf(v);
          v = gelu_erf(v);
        } else if (a.act == MMDTI_ACT_GELU_BWD) {
          v *= gelu_erf_grad(bf2f(a.aux_in[(long long)row * a.ld_aux + col]));
        }
        if (a.drop_thresh) {
          bool keep = dropout_keep(a.seed, a.site, (uint64_t)row * (uint64_t)a.N + col, a.drop_thresh);
          v = keep ? v * a.drop_scale : 0.f;
        }
        if (a.residual && lead) v += a.residual[(long long)row * a.ldr + col];
        const long long ci = coff + (long long)row * a.ldc + col;
        if (a.c_dtype == MMDTI_DT_BF16) {
          reinterpret_cast<bf16_t*>(a.C)[ci] = f2bf(v);
        } else if (a.c_dtype == MMDTI_DT_F32) {
          float* c = reinterpret_cast<float*>(a.C);
          c[ci] = (a.beta != 0.f) ? v + a.beta * c[ci] : v;
        } else {
          atomicAdd(reinterpret_cast<float*>(a.C) + ci, v);
        }
      }
    }
  }
}

}  // namespace mmdti

using namespace mmdti;

extern "C" int mmdti_gemm_bf16(mmdti_stream_t stream, const void* A, const void* B, void* C, int M, int N, int K,
                               int lda, int ldb, int ldc, int transA, int transB, int batch_outer, int batch_inner,
                               long long sAo, long long sAi, long long sBo, long long sBi, long long sCo,
                               long long sCi, int splitk, float alpha, float beta, const float* bias,
                               const float* residual, int ldr, int act, const void* aux_in, void* aux_out,
                               int ld_aux, int c_dtype, float drop_p, unsigned long long seed, unsigned int site) {
  MMDTI_REQUIRE(M > 0 && N > 0 && K > 0, "gemm: M,N,K must be positive (got %d,%d,%d)", M, N, K);
  MMDTI_REQUIRE(A && B && C, "gemm: null operand");
  MMDTI_REQUIRE(aligned16(A) && aligned16(B), "gemm: A and B must be 16-byte aligned");
  MMDTI_REQUIRE(lda % 8 == 0 && ldb % 8 == 0, "gemm: lda/ldb must be multiples of 8 elements (got %d,%d)", lda, ldb);
  MMDTI_REQUIRE(sAo % 8 == 0 && sAi % 8 == 0 && sBo % 8 == 0 && sBi % 8 == 0, "gemm: batch strides must be multiples of 8");
  MMDTI_REQUIRE(batch_outer >= 1 && batch_inner >= 1 && splitk >= 1, "gemm: batch/splitk must be >= 1");
  MMDTI_REQUIRE(c_dtype == MMDTI_DT_F32 || c_dtype == MMDTI_DT_BF16 || c_dtype == MMDTI_DT_F32_ATOMIC, "gemm: bad c_dtype");
  MMDTI_REQUIRE(splitk == 1 || c_dtype == MMDTI_DT_F32_ATOMIC, "gemm: splitk>1 needs the atomic fp32 output mode");
  MMDTI_REQUIRE(splitk == 1 || (act == MMDTI_ACT_NONE && drop_p == 0.f), "gemm: splitk>1 cannot fuse act/dropout");
  MMDTI_REQUIRE(act != MMDTI_ACT_GELU_BWD || aux_in, "gemm: gelu_bwd needs aux_in");
  MMDTI_REQUIRE(batch_outer * batch_inner == 1 || (!residual && !aux_in && !aux_out && drop_p == 0.f),
                "gemm: residual/aux/dropout epilogues are unbatched only");
  MMDTI_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "gemm: dropout p out of range");
  GemmArgs a;
  a.A = (const bf16_t*)A; a.B = (const bf16_t*)B; a.C = C;
  a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc;
  a.batch_inner = batch_inner;
  a.sAo = sAo; a.sAi = sAi; a.sBo = sBo; a.sBi = sBi; a.sCo = sCo; a.sCi = sCi;
  a.splitk = splitk; a.alpha = alpha; a.beta = beta; a.bias = bias; a.residual = residual; a.ldr = ldr;
  a.act = act; a.aux_in = (const bf16_t*)aux_in; a.aux_out = (bf16_t*)aux_out; a.ld_aux = ld_aux; a.c_dtype = c_dtype;
  a.drop_thresh = dropout_thresh(drop_p); a.drop_scale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  a.seed = seed; a.site = site;
  const int tiles = cdiv(M, BM) * cdiv(N, BN);
  dim3 grid(tiles, 1, batch_outer * batch_inner * splitk), block(256);
  MMDTI_REQUIRE(grid.z <= 65535u, "gemm: batch*splitk too large (%u)", grid.z);
  const size_t smem = 4 * BM * LDT * sizeof(bf16_t);
  hipStream_t s = (hipStream_t)stream;
  // 72 KiB of dynamic LDS (> the 64 KiB default): opt in once per instantiation.
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipError_t e3 = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipError_t e4 = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess || e4 != hipSuccess) {
      set_error("gemm: hipFuncSetAttribute(MaxDynamicSharedMemorySize=%zu) failed", smem);
      return MMDTI_ERR_LAUNCH;
    }
    attr_done = true;
  }
  if (transA) {
    if (transB) hipLaunchKernelGGL((gemm_bf16_kernel<true, true>), grid, block, smem, s, a);
    else hipLaunchKernelGGL((gemm_bf16_kernel<true, false>), grid, block, smem, s, a);
  } else {
    if (transB) hipLaunchKernelGGL((gemm_bf16_kernel<false, true>), grid, block, smem, s, a);
    else hipLaunchKernelGGL((gemm_bf16_kernel<false, false>), grid, block, smem, s, a);
  }
  MMDTI_LAUNCH_CHECK();
  return MMDTI_OK;
}
