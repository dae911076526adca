// Shared device/host helpers for the MM-DTI gfx950 kernels.
// gfx950 only: 64-lane waves, bf16 MFMA, 160 KiB LDS.  No CUDA/portability shims.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#include "../../include/mmdti_hip.h"

namespace mmdti {

typedef uint16_t bf16_t;  // raw bf16 bits in memory

// ---- error plumbing (host) -------------------------------------------------
void set_error(const char* fmt, ...);
#define MMDTI_REQUIRE(cond, ...)                 \
  do {                                           \
    if (!(cond)) {                               \
      ::mmdti::set_error(__VA_ARGS__);           \
      return MMDTI_ERR_INVALID;                  \
    }                                            \
  } while (0)
#define MMDTI_LAUNCH_CHECK()                                              \
  do {                                                                    \
    hipError_t e__ = hipGetLastError();                                   \
    if (e__ != hipSuccess) {                                              \
      ::mmdti::set_error("%s:%d launch failed: %s", __FILE__, __LINE__,   \
                         hipGetErrorString(e__));                         \
      return MMDTI_ERR_LAUNCH;                                            \
    }                                                                     \
  } while (0)

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// ---- bf16 <-> f32 ----------------------------------------------------------
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// round-to-nearest-even; NaN stays NaN, +-inf stays inf.  gfx950 has the conversion in hardware (v_cvt_pk_bf16_f32):
// one instruction instead of the six of the integer formulation.
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
// two fp32 -> one word of fp16 bits, round to nearest even, SATURATING: a value past the fp16 range leaves as +-65504 instead of +-inf
// (an inf in a forward GEMM operand would turn the whole row into NaN at the next product; the reference's AMP run has the same
// exposure and no guard -- this is the safer contract, pinned by tests/test_fp16_mode_gpu.py).  Three instructions per pair
// (v_cvt_pk_f16_f32, v_pk_min_f16, v_pk_max_f16) against one for bf16: the store epilogues are VALU-bound at K = 512.  +-inf clamp
// too, and a NaN does not survive the min / max (IEEE minNum) -- the fp32 residual stream written beside every 16-bit copy keeps it.
__device__ __forceinline__ uint32_t f2h_sat2(float lo, float hi) {
  typedef _Float16 h16x2_t __attribute__((ext_vector_type(2)));
  typedef float f32x2v_t __attribute__((ext_vector_type(2)));
  h16x2_t v = __builtin_convertvector(f32x2v_t{lo, hi}, h16x2_t);
  const h16x2_t mx = {(_Float16)65504.f, (_Float16)65504.f};
  v = __builtin_elementwise_min(v, mx);
  v = __builtin_elementwise_max(v, -mx);
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ uint16_t f2h_sat(float f) { return (uint16_t)(f2h_sat2(f, f) & 0xffffu); }
// eight fp16 values (one 16-byte chunk) -> their eight bf16 roundings (fp16 -> fp32 is exact, then round to nearest even):
// element for element what mmdti_cast_f16_bf16 writes
__device__ __forceinline__ uint4 h2bf8(const uint4& u) {
  typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
  typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
  const f16x8_t h = __builtin_bit_cast(f16x8_t, u);
  bf16x8_t o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (__bf16)(float)h[e];
  return __builtin_bit_cast(uint4, o);
}

// ---- wave-level reductions (64 lanes) on the DPP crossbar -------------------------------------------------------
// v_*_dpp reads a neighbour lane as part of a normal VALU op (~1 issue slot) whereas __shfl_xor lowers to ds_bpermute
// (an LDS round trip per step).  Row = 16 lanes.  Prefix-style: after row_shr 1,2,4,8 lane 15 of every row holds the
// row total, row_bcast:15 / row_bcast:31 fold the four rows into lane 63, v_readlane broadcasts it through an SGPR.
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf, bool BOUND_ZERO = true>
__device__ __forceinline__ float dpp_f32(float old, float src) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(src), CTRL, ROW_MASK, BANK_MASK, BOUND_ZERO));
}
constexpr int DPP_ROW_SHL = 0x100, DPP_ROW_SHR = 0x110, DPP_ROW_ROR = 0x120, DPP_ROW_BCAST15 = 0x142, DPP_ROW_BCAST31 = 0x143;
constexpr int DPP_QUAD_XOR1 = 0xB1, DPP_QUAD_XOR2 = 0x4E;  // quad_perm [1,0,3,2] / [2,3,0,1]

__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_f32<DPP_ROW_SHR + 1>(0.f, v);
  v += dpp_f32<DPP_ROW_SHR + 2>(0.f, v);
  v += dpp_f32<DPP_ROW_SHR + 4>(0.f, v);
  v += dpp_f32<DPP_ROW_SHR + 8>(0.f, v);
  v += dpp_f32<DPP_ROW_BCAST15, 0xa>(0.f, v);
  v += dpp_f32<DPP_ROW_BCAST31, 0xc>(0.f, v);
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
  const float NI = -INFINITY;  // lanes without a source keep -inf (bound_ctrl off)
  v = fmaxf(v, dpp_f32<DPP_ROW_SHR + 1, 0xf, 0xf, false>(NI, v));
  v = fmaxf(v, dpp_f32<DPP_ROW_SHR + 2, 0xf, 0xf, false>(NI, v));
  v = fmaxf(v, dpp_f32<DPP_ROW_SHR + 4, 0xf, 0xf, false>(NI, v));
  v = fmaxf(v, dpp_f32<DPP_ROW_SHR + 8, 0xf, 0xf, false>(NI, v));
  v = fmaxf(v, dpp_f32<DPP_ROW_BCAST15, 0xa, 0xf, false>(NI, v));
  v = fmaxf(v, dpp_f32<DPP_ROW_BCAST31, 0xc, 0xf, false>(NI, v));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
// Sum 8 per-lane values over the 64 lanes; every lane ends up holding the total of v[lane & 7].
// Reduce-scatter: each exchange step halves the number of live values (xor 1, 2 by quad_perm, xor 4 by row_shl/shr:4,
// xor 8 by row_ror:8), the last two (xor 16, 32) cross rows and go through ds_bpermute.
__device__ __forceinline__ float wave_sum8_scatter(const float (&v)[8], int lane) {
  const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4;
  float a[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float keep = b0 ? v[2 * i + 1] : v[2 * i];
    const float send = b0 ? v[2 * i] : v[2 * i + 1];
    a[i] = keep + dpp_f32<DPP_QUAD_XOR1>(0.f, send);  // partial of v[2i + b0]
  }
  float b[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const float keep = b1 ? a[2 * i + 1] : a[2 * i];
    const float send = b1 ? a[2 * i] : a[2 * i + 1];
    b[i] = keep + dpp_f32<DPP_QUAD_XOR2>(0.f, send);  // partial of v[4i + 2*b1 + b0]
  }
  const float keep = b2 ? b[1] : b[0];
  const float send = b2 ? b[0] : b[1];
  const float up = dpp_f32<DPP_ROW_SHL + 4>(0.f, send);  // lane l <- lane l+4
  const float dn = dpp_f32<DPP_ROW_SHR + 4>(0.f, send);  // lane l <- lane l-4
  float c = keep + (b2 ? dn : up);                       // partial of v[lane & 7] over this 8-lane group
  c += dpp_f32<DPP_ROW_ROR + 8>(0.f, c);                 // lane l <- lane l^8 (rotate the 16-lane row by 8)
  c += __shfl_xor(c, 16, 64);
  c += __shfl_xor(c, 32, 64);
  return c;
}

// ---- counter-based RNG for dropout ------------------------------------------------------------------------------
// Stateless: 4 x 32 random bits for counter `ctr` under (seed, site).  Built from the lowbias32 avalanche hash
// (2 multiplies + 3 xor-shifts per word); ~10 integer ops per random word, vs ~25 for Philox4x32-10, which made the
// attention-probability dropout (277 M elements per layer at the bench shape) VALU-bound.  Masks only need to be
// unbiased, decorrelated across elements/sites/steps and reproducible from (seed, site, element) -- all tested.
struct Rand4 {
  uint32_t x, y, z, w;
};
// Per-step salt of every dropout stream.  A launch receives (seed, site) by VALUE; when the step is replayed from a captured
// HIP graph those values are frozen into the graph, so the part that must change from step to step lives in device memory:
// one 64-bit word per translation unit (a plain `static __device__` -- no relocatable device code), refreshed from the
// trainer's step-state buffer by mmdti_seed_salt_pull at the top of every step (a graph node like any other).  It stays 0
// unless a caller opts in, and then forward and backward of a step still see the same value.
static __device__ unsigned long long g_seed_salt = 0ull;
__device__ __forceinline__ uint64_t salted(uint64_t seed) { return seed ^ g_seed_salt; }
#define MMDTI_DEFINE_SALT_PULL(TU)                                                                                   \
  namespace mmdti {                                                                                                  \
  __global__ void seed_salt_pull_kernel_##TU(const unsigned long long* __restrict__ src) { g_seed_salt = *src; }     \
  void salt_pull_##TU(hipStream_t s, const unsigned long long* src) {                                                \
    hipLaunchKernelGGL(seed_salt_pull_kernel_##TU, dim3(1), dim3(1), 0, s, src);                                     \
  }                                                                                                                  \
  }
void salt_pull_gemm(hipStream_t s, const unsigned long long* src);
void salt_pull_layernorm(hipStream_t s, const unsigned long long* src);
void salt_pull_pair_attn(hipStream_t s, const unsigned long long* src);
void salt_pull_pair_attn_bwd(hipStream_t s, const unsigned long long* src);
void salt_pull_pair_attn_bwd_g16(hipStream_t s, const unsigned long long* src);
void salt_pull_attn(hipStream_t s, const unsigned long long* src);
void salt_pull_elementwise(hipStream_t s, const unsigned long long* src);
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
// Four uniforms for counter `ctr` from TWO avalanche hashes: each is 16 random bits in the HIGH half of a word, so the
// callers' `r >= thresh` against the 32-bit threshold p*2^32 keeps working (p effectively quantised to 1/65536).  5 -> 2
// hashes per 4 elements: this runs in every dropout site of the GEMM epilogues and the LayerNorm / cast kernels.
__device__ __forceinline__ Rand4 philox4(uint64_t seed, uint32_t site, uint64_t ctr) {
  seed = salted(seed);
  const uint32_t k = mix32((uint32_t)seed ^ (site * 0x9E3779B9u)) ^ (uint32_t)(seed >> 32);  // wave-uniform: hoisted to SALU
  const uint32_t a = mix32(((uint32_t)ctr ^ k) + (uint32_t)(ctr >> 32) * 0x85EBCA6Bu);
  const uint32_t b = mix32(a ^ 0xB5297A4Du);
  return Rand4{a << 16, a & 0xffff0000u, b << 16, b & 0xffff0000u};
}
// keep-mask for element index `idx` of dropout site `site`: one Philox call covers 4 consecutive
// elements (idx>>2), lane picks word idx&3.  keep iff u32 >= thresh, thresh = p * 2^32.
__device__ __forceinline__ bool dropout_keep(uint64_t seed, uint32_t site, uint64_t idx, uint32_t thresh) {
  Rand4 r = philox4(seed, site, idx >> 2);
  uint32_t w = (idx & 3) == 0 ? r.x : (idx & 3) == 1 ? r.y : (idx & 3) == 2 ? r.z : r.w;
  return w >= thresh;
}
// ---- the attention-probability dropout generator (fused BERT attention and pair attention: 277-537 M decisions per layer) ----
//   * per PLANE (one (sequence or molecule, head)): a two-word key from the full avalanche of (seed, salt, site, plane) --
//     wave-uniform, scalar registers;
//   * per QUAD (four consecutive keys of one query; counter < 2^23): ONE 32-bit word from two FULL-RATE 24-bit multiplies,
//     h = mul24(ctr ^ k1, C1); h ^= (h >> 12) ^ k2; h = mul24(h, C2); h ^= h >> 12 -- its four bytes are the four uniforms.
//     (Before: two lowbias32 avalanches = four quarter-rate 32-bit multiplies per quad.  The statistics are pinned by
//     tests/test_kernels_gpu.py::test_attn_dropout_mask_statistics on masks recovered from the kernels; on the CPU restatement
//     the same battery rates the generator with numpy's PCG64, scratch/rng_study.py.)
//   * an element is dropped iff its byte < t8, and t8 is drawn PER QUERY ROW: floor(256 p) + Bernoulli(frac(256 p)) from one
//     more word (counter = row | 2^23) -- P(drop) = p to 2^-24, not p rounded to 1/256.  (Rows' thresholds differ by one count:
//     the drops of one row are correlated at 4e-5.)
// thresh8 = (floor(256 p) << 16) | round(65536 frac(256 p)); 0 = no dropout.
struct Rng24 {
  uint32_t k1, k2;
};
__device__ __forceinline__ Rng24 rng24_key(uint64_t seed, uint32_t site, uint32_t plane) {
  seed = salted(seed);
  const uint32_t a = mix32((mix32((uint32_t)seed ^ (site * 0x9E3779B9u)) ^ (uint32_t)(seed >> 32)) + plane * 0x85EBCA6Bu);
  return Rng24{a, mix32(a ^ 0xB5297A4Du)};
}
__device__ __forceinline__ uint32_t rng24_word(const Rng24& k, uint32_t ctr) {
  uint32_t h = __umul24(ctr ^ k.k1, 0x9E3779u);
  h = h ^ (h >> 12) ^ k.k2;
  h = __umul24(h, 0x85EBCBu);
  return h ^ (h >> 12);
}
__device__ __forceinline__ uint32_t rng24_row_t8(const Rng24& k, uint32_t row, uint32_t thresh8) {
  return (thresh8 >> 16) + ((rng24_word(k, row | 0x800000u) & 0xffffu) < (thresh8 & 0xffffu) ? 1u : 0u);
}
__device__ __forceinline__ bool rng24_kept(uint32_t word, int r, uint32_t t8) { return ((word >> (8 * r)) & 0xffu) >= t8; }
static inline uint32_t dropout_thresh8(float p) {
  if (!(p > 0.f)) return 0u;
  const double x = (double)p * 256.0;
  uint32_t t8 = (uint32_t)x, fr = (uint32_t)((x - (double)t8) * 65536.0 + 0.5);
  if (fr >= 65536u) { t8 += 1; fr = 0; }
  return (t8 << 16) | fr;
}
static inline uint32_t dropout_thresh(float p) {
  if (p <= 0.f) return 0u;
  double t = (double)p * 4294967296.0;
  if (t >= 4294967295.0) return 0xFFFFFFFFu;
  return (uint32_t)t;
}

// ---- Tiled pair planes ("blocked rows").  A (molecule, head) plane of N x N pair values is stored per block of 16 queries; inside a
// block the 4-key groups follow each other, each holding its `vr` query rows x 4 keys (vr = 16, or N - 16 qb in the last block):
//     off(query i, key j) = 16 qb N4 + vr (j & ~3) + 4 (i & 15) + (j & 3),      qb = i >> 4, N4 = N rounded up to 4
// A 16 x 16 tile of a complete block is therefore 256 contiguous elements in MFMA accumulator order (lane = (key group, query), 4
// consecutive keys per lane) -- what the pair-attention kernels stream with one 8- / 16-byte access per lane -- and NOTHING is stored for
// queries or 4-key groups past N: at the reference's 128 atoms + BOS / EOS (N = 130) the planes hold 130 x 132 slots instead of the
// 144 x 144 of whole 16 x 16 tiles (17 % less memory; the traffic was already close to it -- the pad lanes of edge tiles are
// predicated off).  Keys N .. N4 - 1 of a real query are pad slots: -inf
// in S, 0 in G, written by the pair-bias kernel and preserved by every store.  Plane stride: pair_plane(N) elements (a multiple of 8,
// so that planes of 2-byte elements stay 16-byte aligned).
__host__ __device__ __forceinline__ int pair_n4(int N) { return (N + 3) & ~3; }
__host__ __device__ __forceinline__ long long pair_plane(int N) { return ((long long)N * pair_n4(N) + 7) & ~7ll; }
__host__ __device__ __forceinline__ long long pair_off(int N, int i, int j) {
  const int q0 = i & ~15, vr = N - q0 < 16 ? N - q0 : 16;
  return (long long)q0 * pair_n4(N) + vr * (j & ~3) + ((i & 15) << 2) + (j & 3);
}

// erf via Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, i.e. fp32-exact for GELU purposes): one v_exp + one v_rcp + 6
// FMAs instead of ocml's branchy erff (~40 instructions) -- the GELU epilogue was costing 40 % of the fc1 GEMM.
__device__ __forceinline__ float fast_erf(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);   // v_rcp_f32 (1 ulp); __frcp_rn expands to a 10-instruction IEEE divide
  const float poly = ((((1.061405429f * t - 1.453152027f) * t + 1.421413741f) * t - 0.284496736f) * t + 0.254829592f) * t;
  const float r = 1.0f - poly * __expf(-ax * ax);
  return copysignf(r, x);
}
__device__ __forceinline__ float gelu_erf(float x) { return x * 0.5f * (1.0f + fast_erf(x * 0.70710678118654752440f)); }
// gelu'(x) = Phi(x) + x*phi(x); the Gaussian exp(-x^2/2) is the same exponential the erf approximation evaluates
__device__ __forceinline__ float gelu_erf_grad(float x) {
  const float kInvSqrt2Pi = 0.39894228040143267794f;
  const float ax = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);
  const float poly = ((((1.061405429f * t - 1.453152027f) * t + 1.421413741f) * t - 0.284496736f) * t + 0.254829592f) * t;
  const float e = __expf(-ax * ax);
  const float erfv = copysignf(1.0f - poly * e, x);
  return 0.5f * (1.0f + erfv) + x * kInvSqrt2Pi * e;
}

// XCD-aware bijective remap of a linear workgroup id (cdna guide T1): hardware workgroup ids go round-robin over the 8 XCDs
// (each with its own L2), so id -> chunk index gives every XCD a CONTIGUOUS range of the work list -- neighbours in the list
// (tiles of one row block, heads of one molecule) then share an L2 instead of each pulling the same lines into eight.
__device__ __forceinline__ int xcd_chunk(int orig, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
}

// gelu(x) and gelu'(x) from one erf / one exponential (scalar form)
__device__ __forceinline__ void gelu_erf_both(float x, float& y, float& dy) {
  const float kInvSqrt2Pi = 0.39894228040143267794f;
  const float ax = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);
  const float poly = ((((1.061405429f * t - 1.453152027f) * t + 1.421413741f) * t - 0.284496736f) * t + 0.254829592f) * t;
  const float e = __expf(-ax * ax);
  const float phi = 0.5f * (1.0f + copysignf(1.0f - poly * e, x));
  y = x * phi;
  dy = phi + x * kInvSqrt2Pi * e;
}

// Two elements at a time on <2 x float>: the multiplies / fused multiply-adds of the polynomial compile to the packed
// v_pk_mul_f32 / v_pk_fma_f32 forms (two fp32 lanes per instruction); only the reciprocal and the exponential stay
// one-wide.  Same formula, same constants, same results as the scalar functions above.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2_t erf_poly2(f32x2_t ax, f32x2_t& e) {
  const f32x2_t d = 1.0f + 0.3275911f * ax;
  f32x2_t t;
  t[0] = __builtin_amdgcn_rcpf(d[0]);
  t[1] = __builtin_amdgcn_rcpf(d[1]);
  const f32x2_t poly = ((((1.061405429f * t - 1.453152027f) * t + 1.421413741f) * t - 0.284496736f) * t + 0.254829592f) * t;
  const f32x2_t q = -ax * ax;
  e[0] = __expf(q[0]);
  e[1] = __expf(q[1]);
  return 1.0f - poly * e;       // erf(|x|)
}
__device__ __forceinline__ f32x2_t gelu_erf2(f32x2_t x) {
  f32x2_t ax, e;
  ax[0] = fabsf(x[0]); ax[1] = fabsf(x[1]);
  ax *= 0.70710678118654752440f;
  const f32x2_t r = erf_poly2(ax, e);
  f32x2_t er;
  er[0] = copysignf(r[0], x[0]); er[1] = copysignf(r[1], x[1]);
  return x * 0.5f * (1.0f + er);
}
// gelu(x) and gelu'(x) from one erf / one exponential
__device__ __forceinline__ void gelu_erf_both2(f32x2_t x, f32x2_t& y, f32x2_t& dy) {
  const float kInvSqrt2Pi = 0.39894228040143267794f;
  f32x2_t ax, e;
  ax[0] = fabsf(x[0]); ax[1] = fabsf(x[1]);
  ax *= 0.70710678118654752440f;
  const f32x2_t r = erf_poly2(ax, e);
  f32x2_t er;
  er[0] = copysignf(r[0], x[0]); er[1] = copysignf(r[1], x[1]);
  const f32x2_t phi = 0.5f * (1.0f + er);
  y = x * phi;
  dy = phi + x * kInvSqrt2Pi * e;
}
__device__ __forceinline__ f32x2_t gelu_erf_grad2(f32x2_t x) {
  const float kInvSqrt2Pi = 0.39894228040143267794f;
  f32x2_t ax, e;
  ax[0] = fabsf(x[0]); ax[1] = fabsf(x[1]);
  ax *= 0.70710678118654752440f;
  const f32x2_t r = erf_poly2(ax, e);
  f32x2_t er;
  er[0] = copysignf(r[0], x[0]); er[1] = copysignf(r[1], x[1]);
  return 0.5f * (1.0f + er) + x * kInvSqrt2Pi * e;
}

}  // namespace mmdti
