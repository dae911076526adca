// pair_attn.h -- shared by pair_attn.hip (forward) and pair_attn_bwd.hip (backward): element types of the pair tensors and their
// load / store / rounding helpers, the bf16 hi + lo split, the ragged key-tile dispatch.  (Two translation units so that the ~200
// kernel instantiations compile in parallel.)
#pragma once
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
// The forward kernel also exists for q | k | v stored as fp16 (F16: the opt-in fp16 forward-operand mode -- three more mantissa
// bits in the operands of q.k^T and P.v; the raw 16-bit LDS images are the same, only the MFMA and the split of P change)
typedef _Float16 pa_h16x4 __attribute__((ext_vector_type(4)));
template <bool F16>
__device__ __forceinline__ f32x4 pa_mfma16(const pa_s16x4& a, const pa_s16x4& b, const f32x4& c) {
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(pa_h16x4, a), __builtin_bit_cast(pa_h16x4, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
}
// x = hi + lo in the operand type of the product (bf16, or fp16 when F16: 22 mantissa bits kept of the fp32 factor)
template <bool F16>
__device__ __forceinline__ void pa_split4_t(const f32x4& v, pa_s16x4& hi, pa_s16x4& lo) {
  if constexpr (F16) {
    pa_h16x4 h, l;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      h[r] = (_Float16)v[r];
      l[r] = (_Float16)(v[r] - (float)h[r]);
    }
    hi = __builtin_bit_cast(pa_s16x4, h);
    lo = __builtin_bit_cast(pa_s16x4, l);
  } else {
    pa_split4(v, hi, lo);
  }
}
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

}  // namespace mmdti

// key tiles of 16 the MFMA kernels are instantiated for: 17 covers the reference's crop (max_atoms = 256 -> N <= 258, data/conformer.py:53,199-204)
#define PA_MAX_NT 17

static inline int check_common(const char* fn, int B, int N, int H, int ld) {
  MMDTI_REQUIRE(B > 0 && N > 0 && H > 0, "%s: B,N,H must be positive", fn);
  MMDTI_REQUIRE(N <= 320, "%s: N=%d exceeds the supported 320 atoms (+BOS/EOS)", fn, N);
  MMDTI_REQUIRE(ld >= N, "%s: ld (%d) < N (%d)", fn, ld, N);
  MMDTI_REQUIRE((long long)B * H <= 2147483647LL, "%s: grid too large", fn);
  return MMDTI_OK;
}

