// pair_attn_bwd_mfma.h -- the MFMA backward kernel of the pair-bias attention and the launcher of its hot path (compact planes:
// fp16 logits, one instantiation per tile count, dense and ragged).  A header because the hot path is compiled TWICE -- gradient
// chain G in fp32 (pair_attn_bwd.hip) and in bf16 (pair_attn_bwd_g16.hip) -- as two translation units that build in parallel
// (34 instantiations of this kernel each: two minutes of hipcc apiece).
#pragma once
#include "pair_attn.h"

namespace mmdti {

// RAG: see the forward kernel.  Skipped key tiles contribute nothing (P = 0, G = 0) and their G is NOT written: the caller
// hands in a zero-initialised G when the batch is ragged.
template <int NT, bool TILED, bool FULL, int NW, bool RAG, typename ST, typename GT>
__global__ __launch_bounds__(64 * NW, NT > 9 ? ((NT >= 16 && NW == 3) ? 2 : 1) : 3) void pair_attn_bwd_mfma_kernel(const bf16_t* __restrict__ qkv, const ST* __restrict__ s_in,
                                                                 const bf16_t* __restrict__ dO, const GT* __restrict__ gin, GT* __restrict__ gout,
                                                                 bf16_t* __restrict__ dqkv, int N, int H, int ld, float scale,
                                                                 int g_in_zero, uint32_t thresh, float dscale, uint64_t seed,
                                                                 uint32_t site, const int* __restrict__ key_tiles,
                                                                 const int* __restrict__ row_off, int qkv_f16) {
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
  // (packed token rows: see the forward kernel)
  const bool packed = RAG && row_off != nullptr;
  const int row0 = packed ? __builtin_amdgcn_readfirstlane(row_off[b]) : b * N;
  const int rows = packed ? __builtin_amdgcn_readfirstlane(row_off[b + 1]) - row0 : N;
  const bf16_t* base = qkv + (long long)row0 * D3 + h * HD;
  for (int t = tid; t < NP + 2; t += blockDim.x) {
    uint4 q = make_uint4(0u, 0u, 0u, 0u), kk = q, vv = q, dd = q;
    if (t < rows) {
      q = *reinterpret_cast<const uint4*>(base + (long long)t * D3);
      kk = *reinterpret_cast<const uint4*>(base + (long long)t * D3 + D);
      vv = *reinterpret_cast<const uint4*>(base + (long long)t * D3 + 2 * D);
      dd = *reinterpret_cast<const uint4*>(dO + ((long long)row0 + t) * D + h * HD);
      // (fp16 forward operands: the forward multiplied the fp16 q | k | v; the backward's products take their bf16 rounding -- the
      //  values mmdti_cast_f16_bf16 would write -- converted here, on the way into LDS, instead of in a pass over HBM)
      if (qkv_f16) { q = h2bf8(q); kk = h2bf8(kk); vv = h2bf8(vv); }
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
  const Rng24 rk = rng24_key(seed, site, (uint32_t)bh);     // dropout: the (molecule, head) plane's key (common.h)
  auto body = [&](int qb, auto edge_c, auto kt_c) {
    constexpr bool EDGE = decltype(edge_c)::value;
    constexpr int KT = decltype(kt_c)::value;   // key tiles this molecule's sweeps cover (NT unless RAG)
    const int qi = qb * 16 + c16;
    const uint32_t t8 = thresh ? rng24_row_t8(rk, (uint32_t)qi, thresh) : 0u;   // this query row's drop threshold
    const bool qvalid = EDGE ? qi < rows : true;
    const long long rowoff = ((long long)bh * N + (qvalid ? qi : 0)) * ld;   // (also the dropout counter base)
    const int vr = EDGE ? min(16, N - qb * 16) : 16;       // (tiled planes: the blocked rows of common.h, see the forward kernel)
    const long long tbase = (long long)bh * pair_plane(N) + (long long)qb * 16 * pair_n4(N) + ((4 * g < pair_n4(N) ? g : 0) * vr + (qvalid ? c16 : 0)) * 4;
    const ST* sin_p = s_in + (TILED ? tbase : rowoff);     // (predicate-off lanes read the row's first 16 bytes: in bounds)
    const GT* gin_p = gin + (TILED ? tbase : rowoff);
    GT* gout_p = gout + (TILED ? tbase : rowoff);
    const int TSTEP = TILED ? (EDGE ? vr * 16 : 256) : 16;
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
          const uint32_t kw = rng24_word(rk, ((uint32_t)qi * (uint32_t)ld + t * 16 + 4 * g) >> 2);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const bool kp = rng24_kept(kw, r, t8);
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
      *reinterpret_cast<uint2*>(dqkv + ((long long)row0 + qi) * D3 + h * HD + 4 * g) = pk;
    }
#undef PA_PRED
#undef PA_FAST
  };
  const int nQB = packed ? (rows + 15) >> 4 : nKB;
  auto run = [&](auto kt_c) {
    for (int qb = wave; qb < nQB; qb += nwaves) {
      if (TILED && qb * 16 + 16 <= rows) body(qb, std::false_type{}, kt_c);
      else body(qb, std::true_type{}, kt_c);
    }
  };
  if constexpr (RAG) pa_dispatch_kt<NT, NT>(kt, run);
  else run(std::integral_constant<int, NT>{});
  __syncthreads();
  for (int key = tid; key < rows; key += blockDim.x) {   // (packed: the representative pad row is no key -- its sums are the zeros they started as)
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
    bf16_t* dst = dqkv + ((long long)row0 + key) * D3 + h * HD;
    store8_bf16(dst + D, a);
    store8_bf16(dst + 2 * D, c2);
  }
}


// compact planes (layout 3 / 7): one FULL instantiation per tile count; key_tiles != null: the ragged (RAG) sweeps
template <typename GT>
static inline void pa_bwd_compact_launch(int nqb, dim3 grid, dim3 blk, hipStream_t st, const void* qkv_bf16, const void* s, const void* do_bf16,
                                         void* g, void* dqkv_bf16, int N, int H, int ld, float scale, int g_in_zero, uint32_t th8, float sc,
                                         unsigned long long seed, unsigned int site, const int* key_tiles, const int* row_off, int qkv_f16) {
#define PA_MBH(NT, NWV, RG)                                                                                                   \
  hipLaunchKernelGGL((pair_attn_bwd_mfma_kernel<NT, true, true, NWV, RG, _Float16, GT>), grid, blk, 0, st, (const bf16_t*)qkv_bf16, \
                     (const _Float16*)s, (const bf16_t*)do_bf16, (const GT*)g, (GT*)g, (bf16_t*)dqkv_bf16, N, H, ld, scale,    \
                     g_in_zero, th8, sc, (uint64_t)seed, (uint32_t)site, key_tiles, row_off, qkv_f16)
#define PA_MBC(NT)                                                                                   \
  case NT:                                                                                           \
    if (key_tiles) PA_MBH(NT, ((NT % 3 == 0 || NT >= 16 || NT == 5) ? 3 : 4), true);                 \
    else PA_MBH(NT, ((NT % 3 == 0 || NT >= 16 || NT == 5) ? 3 : 4), false);                          \
    break
  switch (nqb) {
    PA_MBC(1); PA_MBC(2); PA_MBC(3); PA_MBC(4); PA_MBC(5); PA_MBC(6); PA_MBC(7); PA_MBC(8); PA_MBC(9); PA_MBC(10); PA_MBC(11);
    PA_MBC(12); PA_MBC(13); PA_MBC(14); PA_MBC(15); PA_MBC(16); PA_MBC(17);
  }
#undef PA_MBC
#undef PA_MBH
}
// the bf16-gradient build of the same launcher (pair_attn_bwd_g16.hip)
void pa_bwd_compact_launch_g16(int nqb, dim3 grid, dim3 blk, hipStream_t st, const void* qkv_bf16, const void* s, const void* do_bf16, void* g,
                               void* dqkv_bf16, int N, int H, int ld, float scale, int g_in_zero, uint32_t th8, float sc,
                               unsigned long long seed, unsigned int site, const int* key_tiles, const int* row_off, int qkv_f16);

}  // namespace mmdti
