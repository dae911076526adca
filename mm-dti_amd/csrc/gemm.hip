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
// an operand stored k-major (transposed) keeps its order in LDS ([k][row] image, 16-byte writes) and is
// transposed for free by ds_read_b64_tr_b16 on the way to the MFMA.
// XCD-aware block remap keeps tiles that share an A row-panel on one XCD's L2.
#include <stdlib.h>
#include <string.h>

#include "common.h"

namespace mmdti {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// One v_mfma_f32_16x16x32 on raw 16-bit operand fragments: bf16 (the step's contract) or fp16 (F16: the opt-in forward-operand
// mode -- the reference's own AMP dtype, tasks/trainer.py:181-182, three more mantissa bits at the same matrix-pipe rate).
template <bool F16>
__device__ __forceinline__ f32x4 mfma32(const bf16x8& b, const bf16x8& a, const f32x4& c) {
  if constexpr (F16) {
    typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, b), __builtin_bit_cast(f16x8, a), c, 0, 0, 0);
  } else {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, c, 0, 0, 0);
  }
}
__device__ __forceinline__ uint16_t f2h(float f) { return f2h_sat(f); }
// two fp32 -> one 32-bit word of bf16 (f16 == 0) or fp16 values
__device__ __forceinline__ uint32_t pack16x2(int f16, float lo, float hi) {
  return f16 ? f2h_sat2(lo, hi) : ((uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16));
}

// A fragment of 8 fp16 values as the 8 bf16 values of their round-to-nearest-even conversion (fp16 -> fp32 is exact, then
// v_cvt_pk_bf16_f32): what mmdti_cast_f16_bf16 writes, element for element.  Default precision mode (fp16 forward operands): a saved
// forward activation is the B operand of a weight-gradient GEMM whose A operand -- an activation gradient -- needs bf16's range, so the
// fp16 tile is fetched as it lies in memory and converted on its way from LDS to the matrix pipe instead of in a pass over HBM.
__device__ __forceinline__ bf16x8 frag_h2bf(const bf16x8& v) {
  typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
  const f16x8 h = __builtin_bit_cast(f16x8, v);
  bf16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (__bf16)(float)h[e];
  return o;
}
template <bool CVT>
__device__ __forceinline__ bf16x8 frag_cvt(const bf16x8& v) {
  if constexpr (CVT) return frag_h2bf(v);
  else return v;
}

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int LDT = BK;      // [row][k] image: unpadded 128-B rows, 16-B chunks XOR-swizzled by (row & 7) -> conflict-free ds_read_b128
constexpr int LDTR = BM;     // [k][row] image: unpadded 256-B rows, 8-B units XOR-swizzled by k -> conflict-free ds_read_b64_tr_b16

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
  int c_f16;              // with MMDTI_DT_BF16: the 16-bit output is fp16 (MMDTI_DT_F16 at the C ABI)
  uint32_t drop_thresh;
  float drop_scale;
  uint64_t seed;
  uint32_t site;
  int vec_ok;             // host-checked: every pointer/stride the vector epilogue touches is 16-byte friendly
  float* colsum;          // [N] fp32 or null: += column sums of the stored C (vector epilogue only) -- a Linear's bias gradient
  int stream_c;           // host-set: C is written with streaming (non-temporal) stores -- outputs too large to be of use in the caches
  int dbg;                // measurement switch (see g_gemm_dbg)
  long long slab;         // split-K partials: split ks writes C + ks * slab (elements) instead of accumulating into C; 0 = off
  const void* zeros;      // >= 16 zero bytes in device memory: source of the K-tail rows of gemm_big_kernel's last K-tile (k-major operands)
  float* arowsum;         // [M] fp32 or null: += sum_k op(A)[m][k] -- for a weight gradient dW = dy^T.x (A = dy, k-major) that
                          // is the Linear's bias gradient; taken with one extra MFMA per A fragment against a ones operand
};

// sum over k of the A rows a wave multiplies: D = ones . fa^T puts rowsum(A[m]) in every register of lane m's column
__device__ __forceinline__ bf16x8 ones_frag() {
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const s16x8 o = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
  return __builtin_bit_cast(bf16x8, o);
}
__device__ __forceinline__ void arowsum_flush(const GemmArgs& a, const f32x4& r0, const f32x4& r1, const f32x4& r2, const f32x4& r3, int row0, int lane) {
  if (lane < 16) {
    const int m = row0 + lane;
    if (m < a.M) atomicAdd(a.arowsum + m, r0[0]);
    if (m + 16 < a.M) atomicAdd(a.arowsum + m + 16, r1[0]);
    if (m + 32 < a.M) atomicAdd(a.arowsum + m + 32, r2[0]);
    if (m + 48 < a.M) atomicAdd(a.arowsum + m + 48, r3[0]);
  }
}

// Staging registers of one operand tile: 4 x 16-B chunks per thread, as NAMED members (an array here ends up as a
// private-memory alloca in some instantiations and the whole pipeline then runs through scratch).
struct Stage4 {
  uint4 v0, v1, v2, v3;
};

// chunk c (0..1023) of a 128x64 tile:
//  !TR: memory is [rows][k] (k contiguous): chunk = 8 consecutive k of one row.
//   TR: memory is [k][rows] (rows contiguous): chunk = 8 consecutive rows of one k.
template <bool TR>
__device__ __forceinline__ uint4 load_chunk(const bf16_t* __restrict__ base, int ld, int row0, int k0, int rows, int K, int c) {
  uint4 v = make_uint4(0, 0, 0, 0);
  if (!TR) {
    const int row = row0 + (c >> 3), k = k0 + (c & 7) * 8;
    if (row < rows && k < K) v = *reinterpret_cast<const uint4*>(base + (long long)row * ld + k);
  } else {
    const int k = k0 + (c >> 4), row = row0 + (c & 15) * 8;
    if (k < K && row < rows) v = *reinterpret_cast<const uint4*>(base + (long long)k * ld + row);
  }
  return v;
}
template <bool TR>
__device__ __forceinline__ void load_tile(Stage4& r, const bf16_t* __restrict__ base, int ld, int row0, int k0, int rows, int K, int tid) {
  r.v0 = load_chunk<TR>(base, ld, row0, k0, rows, K, tid);
  r.v1 = load_chunk<TR>(base, ld, row0, k0, rows, K, tid + 256);
  r.v2 = load_chunk<TR>(base, ld, row0, k0, rows, K, tid + 512);
  r.v3 = load_chunk<TR>(base, ld, row0, k0, rows, K, tid + 768);
}

// Fast path (K % 64 == 0, no ragged 8-row chunks): per-thread BYTE offsets of the four 16-byte chunks relative to the
// tile's k origin are computed once; rows beyond the matrix are clamped to the last valid row/chunk (they only feed output
// rows/columns that are never stored), so the K loop issues bare loads: wave-uniform base + 32-bit lane offset.
struct Off4 {
  uint32_t o0, o1, o2, o3;
};
template <bool TR>
__device__ __forceinline__ uint32_t tile_offset1(int c, int ld, int row0, int rows) {
  if (!TR) {
    const int row = min(row0 + (c >> 3), rows - 1);
    return (uint32_t)(((long long)row * ld + (c & 7) * 8) * 2);
  } else {
    const int row = min(row0 + (c & 15) * 8, rows - 8);
    return (uint32_t)((((long long)(c >> 4)) * ld + row) * 2);
  }
}
template <bool TR>
__device__ __forceinline__ Off4 tile_offsets(int ld, int row0, int rows, int tid) {
  return Off4{tile_offset1<TR>(tid, ld, row0, rows), tile_offset1<TR>(tid + 256, ld, row0, rows),
              tile_offset1<TR>(tid + 512, ld, row0, rows), tile_offset1<TR>(tid + 768, ld, row0, rows)};
}
__device__ __forceinline__ void load_tile_fast(Stage4& r, const bf16_t* kbase, const Off4& off) {
  const char* b = reinterpret_cast<const char*>(kbase);
  r.v0 = *reinterpret_cast<const uint4*>(b + off.o0);
  r.v1 = *reinterpret_cast<const uint4*>(b + off.o1);
  r.v2 = *reinterpret_cast<const uint4*>(b + off.o2);
  r.v3 = *reinterpret_cast<const uint4*>(b + off.o3);
}

// element-offset XOR for the k-major image: moves 8-byte units (4 elements) by 4*((k&3) | ((k>>3)&1)<<2) units, so the 32
// lanes of a tr-read half-wave (k rows 8g+q and 8(g+1)+q, q = 0..3, four 8-byte units each) hit 32 distinct bank pairs.
__device__ __forceinline__ int tr_swz(int k) { return (((k & 3) | (((k >> 3) & 1) << 2)) << 2) * 4; }

template <bool TR>
__device__ __forceinline__ void store_chunk(const uint4& v, bf16_t* lds, int c) {
  if (!TR) {
    const int row = c >> 3, kc = ((c & 7) ^ (row & 7)) * 8;
    *reinterpret_cast<uint4*>(lds + row * LDT + kc) = v;
  } else {  // k-major image [64][LDTR]: the operand keeps its memory order, fragments come out of ds_read_b64_tr_b16
    const int k = c >> 4, row = (c & 15) * 8;
    *reinterpret_cast<uint4*>(lds + k * LDTR + (row ^ tr_swz(k))) = v;
  }
}
template <bool TR>
__device__ __forceinline__ void store_tile(const Stage4& r, bf16_t* lds, int tid) {
  store_chunk<TR>(r.v0, lds, tid);
  store_chunk<TR>(r.v1, lds, tid + 256);
  store_chunk<TR>(r.v2, lds, tid + 512);
  store_chunk<TR>(r.v3, lds, tid + 768);
}

// One MFMA 16x16x32 operand fragment (8 consecutive k of one row per lane) for the 16-row sub-tile starting at `row0`,
// k-step kk (32 wide).
//  !TR: [row][k] image, one ds_read_b128.
//   TR: [k][row] image, two ds_read_b64_tr_b16: per 16-lane group g the instruction reads a 4(k) x 16(row) block and
//       hands lane i column i -- lane 4q+p supplies the address of k-row q, columns 4p..4p+3 (guide T10; semantics
//       pinned by tests/test_kernels_gpu.py::test_probe_tr_read_semantics).
template <bool TR>
__device__ __forceinline__ bf16x8 load_frag(const bf16_t* img, int row0, int kk, int lane) {
  if (!TR) {
    const int row = row0 + (lane & 15);
    return *reinterpret_cast<const bf16x8*>(img + row * LDT + (((kk * 4 + (lane >> 4)) ^ (row & 7)) * 8));
  } else {
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const int k = kk * 32 + (lane >> 4) * 8 + ((lane & 15) >> 2);
    const int col = row0 + (lane & 3) * 4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + k * LDTR + (col ^ tr_swz(k))));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + (k + 4) * LDTR + (col ^ tr_swz(k + 4))));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
  }
}

__device__ __forceinline__ void epi_elem(const GemmArgs& a, float accv, int row, int col, bool lead, long long coff) {
  if (row >= a.M || col >= a.N) return;
  float v = accv * a.alpha;
  if (a.bias && lead) v += a.bias[col];
  if (a.act == MMDTI_ACT_GELU) {
    if (a.aux_out) a.aux_out[(long long)row * a.ld_aux + col] = f2bf(v);
    v = gelu_erf(v);
  } else if (a.act == MMDTI_ACT_GELU_BWD) {
    v *= gelu_erf_grad(bf2f(a.aux_in[(long long)row * a.ld_aux + col]));
  } else if (a.act == MMDTI_ACT_GELU_G) {
    a.aux_out[(long long)row * a.ld_aux + col] = f2bf(gelu_erf_grad(v));
    v = gelu_erf(v);
  } else if (a.act == MMDTI_ACT_MUL_AUX) {
    v *= bf2f(a.aux_in[(long long)row * a.ld_aux + col]);
  }
  if (a.drop_thresh) {
    const bool keep = dropout_keep(a.seed, a.site, (uint64_t)row * (uint64_t)a.N + col, a.drop_thresh);
    v = keep ? v * a.drop_scale : 0.f;
  }
  if (a.residual && lead) v += a.residual[(long long)row * a.ldr + col];
  const long long ci = coff + (long long)row * a.ldc + col;
  if (a.c_dtype == MMDTI_DT_BF16) {
    reinterpret_cast<bf16_t*>(a.C)[ci] = a.c_f16 ? f2h(v) : f2bf(v);
  } else if (a.c_dtype == MMDTI_DT_F32) {
    float* c = reinterpret_cast<float*>(a.C);
    c[ci] = (a.beta != 0.f) ? v + a.beta * c[ci] : v;
  } else {
    atomicAdd(reinterpret_cast<float*>(a.C) + ci, v);
  }
}

constexpr int LDC_W = 68;  // fp32 row stride of a wave's 16x64 epilogue patch (68 = 4 mod 8: conflict-free C-layout writes)

constexpr int LDC_S = BN + 4;  // fp32 row stride of the epilogue's staging image (528 B)

// 16-byte streaming store: the outputs of these GEMMs are 34-270 MB written once and read by a LATER kernel -- kept out of
// the L2's way (measured: -10...-25 % on the N >= 1536 outputs at 65536 rows, -22 % on the fc1 + GELU epilogue at 33280)
__device__ __forceinline__ void nt_store16(void* dst, const uint4& u) {
  typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
  const u32x4_t uv = {u.x, u.y, u.z, u.w};
  __builtin_nontemporal_store(uv, reinterpret_cast<u32x4_t*>(dst));
}

// The 8 bias values of a thread's column piece -- the same piece in every pass of a tile's epilogue -- are loaded ONCE per tile,
// before the first pass.  Re-read inside every piece they cost +25 us on a 92 us GEMM: vmcnt retires in order on gfx9, loads and
// stores alike, so a load issued after the previous piece's stores waits for those stores to reach memory.
struct EpiBias {
  float4 b0, b1;
};
__device__ __forceinline__ EpiBias epi_bias(const GemmArgs& a, int col, bool lead) {
  EpiBias b;
  b.b0 = b.b1 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (a.bias && lead && col < a.N) {
    b.b0 = *reinterpret_cast<const float4*>(a.bias + col);
    b.b1 = *reinterpret_cast<const float4*>(a.bias + col + 4);
  }
  return b;
}

// Fused epilogue on 8 consecutive columns of one row, read from the LDS staging image.  N % 8 == 0, col % 8 == 0 here.
__device__ __forceinline__ void epilogue_oct(const GemmArgs& a, const float* src, int row, int col, bool lead, long long coff, float* cs,
                                             float4 bias0, float4 bias1) {
  float v[8];
  {
    const float4 lo = *reinterpret_cast<const float4*>(src);
    const float4 hi = *reinterpret_cast<const float4*>(src + 4);
    v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] *= a.alpha;
  if (a.bias && lead) {
    const float4 b0 = bias0, b1 = bias1;
    v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
  }
  if (a.act == MMDTI_ACT_GELU) {
    if (a.aux_out) {
      uint4 u;
      u.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
      u.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
      u.z = (uint32_t)f2bf(v[4]) | ((uint32_t)f2bf(v[5]) << 16);
      u.w = (uint32_t)f2bf(v[6]) | ((uint32_t)f2bf(v[7]) << 16);
      nt_store16(a.aux_out + (long long)row * a.ld_aux + col, u);
    }
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
      const f32x2_t y = gelu_erf2(f32x2_t{v[e], v[e + 1]});
      v[e] = y[0];
      v[e + 1] = y[1];
    }
  } else if (a.act == MMDTI_ACT_GELU_BWD) {
    const uint4 u = *reinterpret_cast<const uint4*>(a.aux_in + (long long)row * a.ld_aux + col);
    const uint32_t w4[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
      const f32x2_t gr = gelu_erf_grad2(f32x2_t{__uint_as_float(w4[e >> 1] << 16), __uint_as_float(w4[e >> 1] & 0xffff0000u)});
      v[e] *= gr[0];
      v[e + 1] *= gr[1];
    }
  } else if (a.act == MMDTI_ACT_GELU_G) {
    float gq[8];
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
      f32x2_t y, dy;
      gelu_erf_both2(f32x2_t{v[e], v[e + 1]}, y, dy);
      v[e] = y[0]; v[e + 1] = y[1];
      gq[e] = dy[0]; gq[e + 1] = dy[1];
    }
    uint4 u;
    u.x = (uint32_t)f2bf(gq[0]) | ((uint32_t)f2bf(gq[1]) << 16);
    u.y = (uint32_t)f2bf(gq[2]) | ((uint32_t)f2bf(gq[3]) << 16);
    u.z = (uint32_t)f2bf(gq[4]) | ((uint32_t)f2bf(gq[5]) << 16);
    u.w = (uint32_t)f2bf(gq[6]) | ((uint32_t)f2bf(gq[7]) << 16);
    nt_store16(a.aux_out + (long long)row * a.ld_aux + col, u);
  } else if (a.act == MMDTI_ACT_MUL_AUX) {
    const uint4 u = *reinterpret_cast<const uint4*>(a.aux_in + (long long)row * a.ld_aux + col);
    const uint32_t w4[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
      v[e] *= __uint_as_float(w4[e >> 1] << 16);
      v[e + 1] *= __uint_as_float(w4[e >> 1] & 0xffff0000u);
    }
  }
  if (a.drop_thresh) {  // N % 8 == 0 on this path, so the 8 elements are Philox counters idx/4 and idx/4+1
    const uint64_t idx = (uint64_t)row * (uint64_t)a.N + col;
    const Rand4 r0 = philox4(a.seed, a.site, idx >> 2), r1 = philox4(a.seed, a.site, (idx >> 2) + 1);
    const uint32_t rw[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = rw[e] >= a.drop_thresh ? v[e] * a.drop_scale : 0.f;
  }
  if (a.residual && lead) {
    const float* rp = a.residual + (long long)row * a.ldr + col;
    const float4 r0 = *reinterpret_cast<const float4*>(rp), r1 = *reinterpret_cast<const float4*>(rp + 4);
    v[0] += r0.x; v[1] += r0.y; v[2] += r0.z; v[3] += r0.w; v[4] += r1.x; v[5] += r1.y; v[6] += r1.z; v[7] += r1.w;
  }
  const long long ci = coff + (long long)row * a.ldc + col;
  if (a.c_dtype == MMDTI_DT_BF16) {
    uint4 u;
    u.x = pack16x2(a.c_f16, v[0], v[1]);
    u.y = pack16x2(a.c_f16, v[2], v[3]);
    u.z = pack16x2(a.c_f16, v[4], v[5]);
    u.w = pack16x2(a.c_f16, v[6], v[7]);
    if (a.stream_c) nt_store16(reinterpret_cast<bf16_t*>(a.C) + ci, u);
    else *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(a.C) + ci) = u;
    if (cs) {   // column sums of the ROUNDED values: what a column-sum pass over C would add up
      cs[0] += __uint_as_float(u.x << 16); cs[1] += __uint_as_float(u.x & 0xffff0000u);
      cs[2] += __uint_as_float(u.y << 16); cs[3] += __uint_as_float(u.y & 0xffff0000u);
      cs[4] += __uint_as_float(u.z << 16); cs[5] += __uint_as_float(u.z & 0xffff0000u);
      cs[6] += __uint_as_float(u.w << 16); cs[7] += __uint_as_float(u.w & 0xffff0000u);
    }
  } else {
    float* c = reinterpret_cast<float*>(a.C) + ci;
    if (a.beta != 0.f) {
      const float4 c0 = *reinterpret_cast<const float4*>(c), c1 = *reinterpret_cast<const float4*>(c + 4);
      v[0] += a.beta * c0.x; v[1] += a.beta * c0.y; v[2] += a.beta * c0.z; v[3] += a.beta * c0.w;
      v[4] += a.beta * c1.x; v[5] += a.beta * c1.y; v[6] += a.beta * c1.z; v[7] += a.beta * c1.w;
    }
    if (a.stream_c) {
      nt_store16(c, make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])));
      nt_store16(c + 4, make_uint4(__float_as_uint(v[4]), __float_as_uint(v[5]), __float_as_uint(v[6]), __float_as_uint(v[7])));
    } else {
      *reinterpret_cast<float4*>(c) = make_float4(v[0], v[1], v[2], v[3]);
      *reinterpret_cast<float4*>(c + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
    if (cs) {
#pragma unroll
      for (int e = 0; e < 8; ++e) cs[e] += v[e];
    }
  }
}

// Column sums of one 128x128 output tile: every thread summed 8 columns (tid & 15) over its 8 rows in the epilogue; fold
// the 16 row groups through LDS, one atomic per column.
__device__ __forceinline__ void epilogue_colsum(const GemmArgs& a, float* lds, const float* cs, int tid, int n0) {
  __syncthreads();   // the staging image has been read out
#pragma unroll
  for (int e = 0; e < 8; ++e) lds[(tid >> 4) * 128 + (tid & 15) * 8 + e] = cs[e];
  __syncthreads();
  if (tid < 128 && n0 + tid < a.N) {
    float t = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) t += lds[j * 128 + tid];
    atomicAdd(a.colsum + n0 + tid, t);
  }
}

// BCVT: B holds fp16 and is converted to bf16 fragment by fragment (frag_h2bf) -- the weight-gradient forms, A = dy (bf16), B = a
// saved forward activation of the fp16 forward-operand mode
template <bool TA, bool TB, bool FAST, bool F16 = false, bool BCVT = false>
__global__ __launch_bounds__(256, 3) void gemm_bf16_kernel(GemmArgs a) {
  // LDS image: [buf 0: A | B][buf 1: A | B]; addressed by integer offsets from ONE __shared__ base so that every access
  // stays a ds_* instruction (pointer arrays indexed at run time decay to flat loads + scratch).
  extern __shared__ __attribute__((aligned(16))) bf16_t smem[];
  constexpr int TILE = BM * LDT;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;

  // XCD-aware remap of the (n-tile, m-tile) index: consecutive hardware block ids go round-robin over the 8 XCDs,
  // so give each XCD a contiguous chunk of the tile list (bijective form, cdna guide T1).
  const int tiles_n = (a.N + BN - 1) / BN, tiles_m = (a.M + BM - 1) / BM;
  const int nwg = tiles_n * tiles_m;
  const int orig = blockIdx.x;
  const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
  const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  const int tm = wg / tiles_n, tn = wg - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  const int z = blockIdx.z;
  const int zb = z / a.splitk, ks = z - zb * a.splitk;
  const int zo = zb / a.batch_inner, zi = zb - zo * a.batch_inner;
  const bf16_t* __restrict__ A = a.A + zo * a.sAo + zi * a.sAi;
  const bf16_t* __restrict__ B = a.B + zo * a.sBo + zi * a.sBi;

  // K range of this split
  const int ktiles = (a.K + BK - 1) / BK;
  const int per = (ktiles + a.splitk - 1) / a.splitk;
  const int kt0 = ks * per, kt1 = min(ktiles, kt0 + per);
  if (kt0 >= kt1) return;  // empty split (uniform per block): nothing to contribute

  f32x4 acc[4][4] = {};

  // Software pipeline: the global loads of tile kt+1 are issued before tile kt is multiplied out of LDS and land in a
  // register set that is written to the other LDS buffer after the MFMAs; one barrier per K-step.  (A two-register-set
  // variant with two tiles in flight measured no faster -- the loop is bound by LDS/issue bandwidth, not load latency --
  // and hipcc turned its ping-pong register sets into scratch; see DESIGN.md.)
  Stage4 ra, rb;
  Off4 offA = {0, 0, 0, 0}, offB = {0, 0, 0, 0};
  // element stride of one K-tile along the operand's k axis
  const long long kstepA = TA ? (long long)BK * a.lda : BK, kstepB = TB ? (long long)BK * a.ldb : BK;
  if (FAST) {
    offA = tile_offsets<TA>(a.lda, m0, a.M, tid);
    offB = tile_offsets<TB>(a.ldb, n0, a.N, tid);
  }
#define GEMM_LOAD(KT)                                                    \
  if (FAST) {                                                            \
    load_tile_fast(ra, A + (KT) * kstepA, offA);                         \
    load_tile_fast(rb, B + (KT) * kstepB, offB);                         \
  } else {                                                               \
    load_tile<TA>(ra, A, a.lda, m0, (KT) * BK, a.M, a.K, tid);           \
    load_tile<TB>(rb, B, a.ldb, n0, (KT) * BK, a.N, a.K, tid);           \
  }
  GEMM_LOAD(kt0);
  store_tile<TA>(ra, smem, tid);
  store_tile<TB>(rb, smem + TILE, tid);
  __syncthreads();

  // (everything that indexes acc[][] is spelled out with literal indices: a loop the optimizer declines to unroll would
  //  turn the accumulators into a scratch array)
// operands swapped (B fragment first): the accumulator then holds C^T tiles, i.e. lane = output ROW (lane&15) and the 4
  // registers = 4 consecutive output COLUMNS (4*(lane>>4)+r) -- the epilogue stores 8/16 contiguous bytes per lane
  // straight from registers, no LDS transpose.
#define MF(I, J) acc[I][J] = mfma32<F16>(fb##J, fa##I, acc[I][J])
#define GEMM_KK(IMGA, IMGB, KK)                                                                    \
  {                                                                                                \
    const bf16x8 fa0 = load_frag<TA>(IMGA, wr * 64 + 0, KK, lane), fa1 = load_frag<TA>(IMGA, wr * 64 + 16, KK, lane), \
                 fa2 = load_frag<TA>(IMGA, wr * 64 + 32, KK, lane), fa3 = load_frag<TA>(IMGA, wr * 64 + 48, KK, lane); \
    const bf16x8 fb0 = frag_cvt<BCVT>(load_frag<TB>(IMGB, wc * 64 + 0, KK, lane)), fb1 = frag_cvt<BCVT>(load_frag<TB>(IMGB, wc * 64 + 16, KK, lane)), \
                 fb2 = frag_cvt<BCVT>(load_frag<TB>(IMGB, wc * 64 + 32, KK, lane)), fb3 = frag_cvt<BCVT>(load_frag<TB>(IMGB, wc * 64 + 48, KK, lane)); \
    MF(0, 0); MF(0, 1); MF(0, 2); MF(0, 3); MF(1, 0); MF(1, 1); MF(1, 2); MF(1, 3);                \
    MF(2, 0); MF(2, 1); MF(2, 2); MF(2, 3); MF(3, 0); MF(3, 1); MF(3, 2); MF(3, 3);                \
  }
  for (int kt = kt0; kt < kt1; ++kt) {
    const bool more = (kt + 1 < kt1);
    // (the explicit zero on the last step keeps the staging registers defined on every path; without it hipcc demotes
    //  them to a private-memory alloca on the bare-load path and the pipeline runs through scratch)
    if (more) {
      GEMM_LOAD(kt + 1);
    } else {
      const uint4 z = make_uint4(0, 0, 0, 0);
      ra = Stage4{z, z, z, z};
      rb = ra;
    }
    GEMM_KK(smem, smem + TILE, 0);
    GEMM_KK(smem, smem + TILE, 1);
    __syncthreads();  // every wave has read tile kt
    if (more) {
      store_tile<TA>(ra, smem, tid);
      store_tile<TB>(rb, smem + TILE, tid);
      __syncthreads();
    }
  }
#undef GEMM_KK
#undef MF
#undef GEMM_LOAD

  // ---- epilogue.  Accumulator tile (I, J) of this wave: lane -> row m0 + wr*64 + I*16 + (lane&15), registers -> columns
  // n0 + wc*64 + J*16 + 4*(lane>>4) + {0..3}.  Vector path: bias / aux / residual arrive as 8- or 16-byte loads and C
  // leaves as one 8-byte (bf16) or 16-byte (fp32) store per lane, straight from registers.  Atomic (split-K) and
  // unaligned outputs go through a private 16 x 68 fp32 LDS patch per wave so that a wave instruction covers 256
  // contiguous bytes of one C row.  Literal accumulator indices throughout (a rolled loop would index acc[][]
  // dynamically and push it to scratch).  The K-loop ends on a barrier, so the patches may overwrite the tiles.
  const long long coff = zo * a.sCo + zi * a.sCi;
  const bool lead = (ks == 0);
  const int g4 = (lane >> 4) * 4, l15 = lane & 15;
  if (a.c_dtype != MMDTI_DT_F32_ATOMIC && a.vec_ok) {
    // two half-tiles of 64 rows through a [64][132] fp32 LDS image (33,792 B -- no more than the K-loop's tiles): the
    // waves of row-half h park their accumulators (16-byte LDS writes), then all 256 threads run the fused epilogue on
    // 8 contiguous columns each: every wave instruction moves full 128-byte lines of C / residual / aux.
    float* sC = reinterpret_cast<float*>(smem);
#define STG_Q(I, J) *reinterpret_cast<f32x4*>(sC + ((I) * 16 + l15) * LDC_S + wc * 64 + (J) * 16 + g4) = acc[I][J]
#define STG_R(I) STG_Q(I, 0); STG_Q(I, 1); STG_Q(I, 2); STG_Q(I, 3)
#define EPI_HALF(H)                                                        \
    if (wr == (H)) { STG_R(0); STG_R(1); STG_R(2); STG_R(3); }             \
    __syncthreads();                                                       \
    for (int it = 0; it < 4; ++it) {                                       \
      const int chunk = tid + it * 256;                                    \
      const int r = chunk >> 4, cc = (chunk & 15) * 8;                     \
      const int row = m0 + (H) * 64 + r, col = n0 + cc;                    \
      if (row < a.M && col < a.N) epilogue_oct(a, sC + r * LDC_S + cc, row, col, lead, coff, a.colsum ? cs : nullptr, ebias.b0, ebias.b1); \
    }
    float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const EpiBias ebias = epi_bias(a, n0 + (tid & 15) * 8, lead);
    EPI_HALF(0)
    __syncthreads();
    EPI_HALF(1)
    if (a.colsum) epilogue_colsum(a, sC, cs, tid, n0);
#undef EPI_HALF
#undef STG_R
#undef STG_Q
    return;
  }
  float* sW = reinterpret_cast<float*>(smem) + wave * (16 * LDC_W);
#define STG_T(I, J) *reinterpret_cast<f32x4*>(sW + l15 * LDC_W + (J) * 16 + g4) = acc[I][J]
#define EPI_PASS(I)                                                                                  \
  {                                                                                                  \
    STG_T(I, 0); STG_T(I, 1); STG_T(I, 2); STG_T(I, 3);                                              \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                                           \
    __builtin_amdgcn_wave_barrier();                                                                 \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");                                           \
    for (int r = 0; r < 16; ++r)                                                                     \
      epi_elem(a, sW[r * LDC_W + lane], m0 + wr * 64 + (I) * 16 + r, n0 + wc * 64 + lane, lead, coff); \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                                           \
    __builtin_amdgcn_wave_barrier();                                                                 \
  }
  EPI_PASS(0) EPI_PASS(1) EPI_PASS(2) EPI_PASS(3)
#undef EPI_PASS
#undef STG_T
}

// =====================================================================================================================
// LDS-DMA variant of the same tile (bare-load shapes only): the (A|B) tile pair is fetched with
// global_load_lds_dwordx4 -- 16 bytes per lane straight into LDS, no staging registers, no ds_write pass.  The DMA writes
// lane i of a wave-instruction at base + 16*i, so the LDS image stays the swizzled one the fragment loaders expect by
// permuting the SOURCE: the lane that owns LDS slot (row, s) fetches global chunk s ^ swizzle(row) (an involution).
// No load/compute overlap inside a workgroup (fetch, barrier, multiply, barrier); with ~40 fewer registers four
// workgroups fit a CU and overlap each other instead.
template <bool TR>
__device__ __forceinline__ uint32_t glds_offset1(int c, int ld, int row0, int rows) {
  if (!TR) {
    const int rl = c >> 3, kc = (c & 7) ^ (rl & 7);
    const int row = min(row0 + rl, rows - 1);
    return (uint32_t)(((long long)row * ld + kc * 8) * 2);
  } else {
    const int k = c >> 4, rs = (c & 15) ^ (tr_swz(k) >> 3);
    const int row = min(row0 + rs * 8, rows - 8);
    return (uint32_t)(((long long)k * ld + row) * 2);
  }
}
template <bool TR>
__device__ __forceinline__ Off4 glds_offsets(int ld, int row0, int rows, int tid) {
  return Off4{glds_offset1<TR>(tid, ld, row0, rows), glds_offset1<TR>(tid + 256, ld, row0, rows),
              glds_offset1<TR>(tid + 512, ld, row0, rows), glds_offset1<TR>(tid + 768, ld, row0, rows)};
}
// LDS-DMA issued through inline assembly, for the kernels that keep DMA in flight ACROSS barriers (gemm_big_*, the deep ring):
// the compiler's wait-count pass cannot tell a ds_read_b64_tr_b16 (an intrinsic: no alias information) from a reader of the
// tile in flight and put `s_waitcnt vmcnt(0)` in front of the first fragment read after every barrier of the k-major
// variants -- the counted waits of those kernels were dead code (round 3 finding, read off the ISA).  A DMA the pass does not
// see leaves the ordering to the kernel: counted s_waitcnt vmcnt(n) + s_barrier before a staged buffer is read, vmcnt(0)
// before LDS is reused by the epilogue.  (Loads the compiler issues itself stay correct: vmcnt retires loads in order, so
// its own counts are at worst conservative with these in flight.)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ int lds_addr_uniform(const void* p) {
  typedef __attribute__((address_space(3))) void lptr_t;
  return __builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(lptr_t*)p);
}
// 16 bytes per lane from (wave-uniform base + per-lane 32-bit byte offset) to LDS at lds + lane * 16
__device__ __forceinline__ void dma16_su(const void* sbase, uint32_t voff, const bf16_t* lds) {
  const int m = lds_addr_uniform(lds);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(m), "v"(voff), "s"(sbase) : "memory", "m0");
}
// the same from a per-lane 64-bit address
__device__ __forceinline__ void dma16_v(const void* vptr, const bf16_t* lds) {
  const int m = lds_addr_uniform(lds);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" : : "s"(m), "v"(vptr) : "memory", "m0");
}
#pragma clang diagnostic pop
// LDS-DMA with everything wave-uniform on the scalar side: the source is a scalar pointer + ONE per-thread 32-bit offset, the LDS
// destination an INTEGER byte address the caller keeps in an SGPR (lane i lands at m0v + 16 i).  Round 4, read off the ISA: handing the
// DMA helpers LDS POINTERS cost, per issue, a generic-to-LDS address cast with its null check and v_readfirstlane moves, the builtin
// form a 64-bit per-lane address add -- 24 (128 x 128 kernels) to 140 (256 x 256 weight gradients) non-essential vector instructions
// per K-tile and wave, in loops whose every VALU instruction shows up in the launch time.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
// (readfirstlane: folded away where the compiler can prove m0v uniform -- the hot loops --, and what keeps the "s" constraint honest
//  where its uniformity analysis gives up)
__device__ __forceinline__ void dma16_m0(const void* sbase, uint32_t voff, int m0v) {
  const int m = __builtin_amdgcn_readfirstlane(m0v);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(m), "v"(voff), "s"(sbase) : "memory", "m0");
}
__device__ __forceinline__ void dma16_m0v(const void* vptr, int m0v) {
  const int m = __builtin_amdgcn_readfirstlane(m0v);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" : : "s"(m), "v"(vptr) : "memory", "m0");
}
#pragma clang diagnostic pop
// one 128 x 64 operand tile: four chunks per thread, 4 KB apart in the image (m0v: the wave's 1 KB slot of the image)
__device__ __forceinline__ void glds_tile_m0(const void* kbase, const Off4& off, int m0v) {
  dma16_m0(kbase, off.o0, m0v);
  dma16_m0(kbase, off.o1, m0v + 4096);
  dma16_m0(kbase, off.o2, m0v + 8192);
  dma16_m0(kbase, off.o3, m0v + 12288);
}
__device__ __forceinline__ void glds_tile_asm(const bf16_t* kbase, const Off4& off, bf16_t* img, int wave) {
  const bf16_t* d = img + wave * 512;
  dma16_su(kbase, off.o0, d);
  dma16_su(kbase, off.o1, d + 2048);
  dma16_su(kbase, off.o2, d + 4096);
  dma16_su(kbase, off.o3, d + 6144);
}
__device__ __forceinline__ void glds_tile(const bf16_t* kbase, const Off4& off, bf16_t* img, int wave) {
  typedef __attribute__((address_space(1))) const void gptr_t;
  typedef __attribute__((address_space(3))) void lptr_t;
  const char* b = reinterpret_cast<const char*>(kbase);
  bf16_t* d = img + wave * 512;   // 64 lanes x 8 elements per wave-instruction; instruction j covers chunks j*256 ...
  __builtin_amdgcn_global_load_lds((gptr_t*)(b + off.o0), (lptr_t*)(d), 16, 0, 0);
  __builtin_amdgcn_global_load_lds((gptr_t*)(b + off.o1), (lptr_t*)(d + 2048), 16, 0, 0);
  __builtin_amdgcn_global_load_lds((gptr_t*)(b + off.o2), (lptr_t*)(d + 4096), 16, 0, 0);
  __builtin_amdgcn_global_load_lds((gptr_t*)(b + off.o3), (lptr_t*)(d + 6144), 16, 0, 0);
}

// DBUF: two (A|B) tile pairs in LDS, the DMA of tile kt+1 in flight while tile kt is multiplied (one barrier per K-step,
// 64 KB -> 2 workgroups per CU): for the long-K split-K weight gradients, where depth of pipeline beats occupancy.
// DBUF == 2 (DEEP): a ring of four (A|B) tile pairs (128 KB: the workgroup owns its CU), three K-steps of DMA in flight
// across raw s_barriers, retired with counted s_waitcnt vmcnt -- for launches of at most one workgroup per CU (small
// batches: 13-56 tiles), where nothing else on the CU hides a K-step's ~1 us fetch latency.  Same products in the same
// order as the other two forms: bit-identical results.
constexpr int DEEP_STAGES = 4;
template <bool TA, bool TB, int DBUF, bool F16 = false, bool BCVT = false>
__global__ __launch_bounds__(256, DBUF == 2 ? 1 : DBUF ? 2 : 4) void gemm_glds_kernel(GemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) bf16_t smem[];
  constexpr int TILE = BM * LDT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int tiles_n = (a.N + BN - 1) / BN, tiles_m = (a.M + BM - 1) / BM;
  const int nwg = tiles_n * tiles_m;
  const int orig = blockIdx.x;
  const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
  const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  const int tm = wg / tiles_n, tn = wg - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int z = blockIdx.z;
  const int zb = z / a.splitk, ks = z - zb * a.splitk;
  const int zo = zb / a.batch_inner, zi = zb - zo * a.batch_inner;
  const bf16_t* __restrict__ A = a.A + zo * a.sAo + zi * a.sAi;
  const bf16_t* __restrict__ B = a.B + zo * a.sBo + zi * a.sBi;
  const int ktiles = a.K / BK;
  const int per = (ktiles + a.splitk - 1) / a.splitk;
  const int kt0 = ks * per, kt1 = min(ktiles, kt0 + per);
  if (kt0 >= kt1) return;
  f32x4 acc[4][4] = {};
  const Off4 offA = glds_offsets<TA>(a.lda, m0, a.M, tid), offB = glds_offsets<TB>(a.ldb, n0, a.N, tid);
  const long long kstepA = TA ? (long long)BK * a.lda : BK, kstepB = TB ? (long long)BK * a.ldb : BK;
#define MF(I, J) acc[I][J] = mfma32<F16>(fb##J, fa##I, acc[I][J])
#define GEMM_KK(IMGA, IMGB, KK)                                                                    \
  {                                                                                                \
    const bf16x8 fa0 = load_frag<TA>(IMGA, wr * 64 + 0, KK, lane), fa1 = load_frag<TA>(IMGA, wr * 64 + 16, KK, lane), \
                 fa2 = load_frag<TA>(IMGA, wr * 64 + 32, KK, lane), fa3 = load_frag<TA>(IMGA, wr * 64 + 48, KK, lane); \
    const bf16x8 fb0 = frag_cvt<BCVT>(load_frag<TB>(IMGB, wc * 64 + 0, KK, lane)), fb1 = frag_cvt<BCVT>(load_frag<TB>(IMGB, wc * 64 + 16, KK, lane)), \
                 fb2 = frag_cvt<BCVT>(load_frag<TB>(IMGB, wc * 64 + 32, KK, lane)), fb3 = frag_cvt<BCVT>(load_frag<TB>(IMGB, wc * 64 + 48, KK, lane)); \
    MF(0, 0); MF(0, 1); MF(0, 2); MF(0, 3); MF(1, 0); MF(1, 1); MF(1, 2); MF(1, 3);                \
    MF(2, 0); MF(2, 1); MF(2, 2); MF(2, 3); MF(3, 0); MF(3, 1); MF(3, 2); MF(3, 3);                \
    if (TA && DBUF == 1 && do_rs) {                                                                \
      rs0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fa0, rs0, 0, 0, 0);                      \
      rs1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fa1, rs1, 0, 0, 0);                      \
      rs2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fa2, rs2, 0, 0, 0);                      \
      rs3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fa3, rs3, 0, 0, 0);                      \
    }                                                                                              \
  }
  const bool do_rs = TA && DBUF == 1 && a.arowsum != nullptr && tn == 0 && wc == 0;   // (double-buffered variant only: register room)
  const bf16x8 ones = ones_frag();
  f32x4 rs0 = {0.f, 0.f, 0.f, 0.f}, rs1 = rs0, rs2 = rs0, rs3 = rs0;
  // LDS byte address of this wave's 1 KB slot of the A image of stage 0 (the B image 16 KB further, a stage's pair 32 KB)
  const int lds_w = lds_addr_uniform(smem) + __builtin_amdgcn_readfirstlane(wave) * 1024;
  constexpr int TB2 = TILE * 2;
  if (!DBUF) {
    const char* pa = reinterpret_cast<const char*>(A + kt0 * kstepA);
    const char* pb = reinterpret_cast<const char*>(B + kt0 * kstepB);
    for (int kt = kt0; kt < kt1; ++kt) {
      glds_tile_m0(pa, offA, lds_w);
      glds_tile_m0(pb, offB, lds_w + TB2);
      pa += kstepA * 2;
      pb += kstepB * 2;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the DMA is issued from inline assembly: the compiler does not count it)
      __syncthreads();   // the DMA'd tile is visible to every wave
      GEMM_KK(smem, smem + TILE, 0);
      GEMM_KK(smem, smem + TILE, 1);
      __syncthreads();   // every wave has read the tile before the next fetch overwrites it
    }
  } else if (DBUF == 2) {
    const int nt = kt1 - kt0;
    const bf16_t* __restrict__ A0 = A + kt0 * kstepA;
    const bf16_t* __restrict__ B0 = B + kt0 * kstepB;
    // (a stage past the end of the K range re-fetches the last tile into a buffer nobody reads again: every thread then
    //  always has the same number of loads in flight, which is what the counted waits assume)
#define DEEP_ISSUE(T)                                                                        \
    {                                                                                        \
      const int tt = min((T), nt - 1);                                                       \
      const int dst = lds_w + ((T) % DEEP_STAGES) * (2 * TB2);                               \
      glds_tile_m0(A0 + tt * kstepA, offA, dst);                                             \
      glds_tile_m0(B0 + tt * kstepB, offB, dst + TB2);                                       \
    }
    DEEP_ISSUE(0); DEEP_ISSUE(1); DEEP_ISSUE(2);
    for (int t = 0; t < nt; ++t) {
      asm volatile("s_waitcnt vmcnt(16)" ::: "memory");   // 8 loads per stage: stages t+1, t+2 may stay in flight
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();   // stage t has landed for every wave; every wave is done reading stage t-1
      __builtin_amdgcn_sched_barrier(0);
      DEEP_ISSUE(t + 3);              // -> the buffer of stage t-1
      __builtin_amdgcn_sched_barrier(0);
      const bf16_t* img = smem + (t % DEEP_STAGES) * (2 * TILE);
      GEMM_KK(img, img + TILE, 0);
      GEMM_KK(img, img + TILE, 1);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();     // no DMA in flight, no read outstanding: LDS is free for the epilogue
#undef DEEP_ISSUE
  } else {
    // (inline-assembly DMA here too: behind the builtin the compiler drained vmcnt in front of the first ds_read_b64_tr_b16 of
    //  tile kt -- the prefetch of tile kt+1 never ran under the multiply on the k-major operands this variant exists for)
    glds_tile_m0(A + kt0 * kstepA, offA, lds_w);
    glds_tile_m0(B + kt0 * kstepB, offB, lds_w + TB2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int cur = 0;
    for (int kt = kt0; kt < kt1; ++kt) {
      const int nxt = lds_w + (cur ^ 1) * (2 * TB2);
      if (kt + 1 < kt1) {
        glds_tile_m0(A + (kt + 1) * kstepA, offA, nxt);
        glds_tile_m0(B + (kt + 1) * kstepB, offB, nxt + TB2);
      }
      __builtin_amdgcn_sched_barrier(0);
      const bf16_t* img = smem + cur * (2 * TILE);
      GEMM_KK(img, img + TILE, 0);
      GEMM_KK(img, img + TILE, 1);
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // the DMA of tile kt+1 has landed, the reads of tile kt are retired
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      cur ^= 1;
    }
  }
#undef GEMM_KK
#undef MF
  if (TA && DBUF == 1 && do_rs) arowsum_flush(a, rs0, rs1, rs2, rs3, m0 + wr * 64, lane);
  const long long coff = zo * a.sCo + zi * a.sCi;
  const bool lead = (ks == 0);
  const int g4 = (lane >> 4) * 4, l15 = lane & 15;
  if (a.c_dtype != MMDTI_DT_F32_ATOMIC && a.vec_ok) {
    float* sC = reinterpret_cast<float*>(smem);
#define STG_Q(I, J) *reinterpret_cast<f32x4*>(sC + ((I) * 16 + l15) * LDC_S + wc * 64 + (J) * 16 + g4) = acc[I][J]
#define STG_R(I) STG_Q(I, 0); STG_Q(I, 1); STG_Q(I, 2); STG_Q(I, 3)
#define EPI_HALF(H)                                                        \
    if (wr == (H)) { STG_R(0); STG_R(1); STG_R(2); STG_R(3); }             \
    __syncthreads();                                                       \
    for (int it = 0; it < 4; ++it) {                                       \
      const int chunk = tid + it * 256;                                    \
      const int rr = chunk >> 4, cc = (chunk & 15) * 8;                    \
      const int row = m0 + (H) * 64 + rr, col = n0 + cc;                   \
      if (row < a.M && col < a.N) epilogue_oct(a, sC + rr * LDC_S + cc, row, col, lead, coff, a.colsum ? cs : nullptr, ebias.b0, ebias.b1); \
    }
    float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const EpiBias ebias = epi_bias(a, n0 + (tid & 15) * 8, lead);
    EPI_HALF(0)
    __syncthreads();
    EPI_HALF(1)
    if (a.colsum) epilogue_colsum(a, sC, cs, tid, n0);
#undef EPI_HALF
#undef STG_R
#undef STG_Q
    return;
  }
  float* sW = reinterpret_cast<float*>(smem) + wave * (16 * LDC_W);
#define STG_T(I, J) *reinterpret_cast<f32x4*>(sW + l15 * LDC_W + (J) * 16 + g4) = acc[I][J]
#define EPI_PASS(I)                                                                                  \
  {                                                                                                  \
    STG_T(I, 0); STG_T(I, 1); STG_T(I, 2); STG_T(I, 3);                                              \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                                           \
    __builtin_amdgcn_wave_barrier();                                                                 \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");                                           \
    for (int rr = 0; rr < 16; ++rr)                                                                  \
      epi_elem(a, sW[rr * LDC_W + lane], m0 + wr * 64 + (I) * 16 + rr, n0 + wc * 64 + lane, lead, coff); \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                                           \
    __builtin_amdgcn_wave_barrier();                                                                 \
  }
  EPI_PASS(0) EPI_PASS(1) EPI_PASS(2) EPI_PASS(3)
#undef EPI_PASS
#undef STG_T
}

// =====================================================================================================================
// 64 x 64 tiles for SMALL launches -- the reference's default batch of 16-32 molecules gives GEMMs of 1-3 k rows: 13-56
// tiles of 128 x 128 on 256 CUs, each a lone workgroup walking 8-32 K-steps whose fetch -> barrier -> fragment reads -> MFMA
// chain nothing overlaps (measured: ~1.45 us per K-step single-buffered, ~0.97 us behind the four-stage ring above).  A
// quarter of the tile per workgroup puts 4 x as many CUs to work and shortens the chain of a K-step to 8 fragment reads +
// 8 MFMAs per wave; the (A|B) pairs arrive through a four-stage LDS-DMA ring (inline-assembly DMA, counted waits: two
// stages stay in flight across the barrier).  A row-major ([M,K]) only; B row-major ([N,K]) or k-major ([K,N]); the
// vector epilogue only (the dispatch keeps everything else on the 128 x 128 kernels).  Products are summed in the same
// order as there: results are bit-identical.
constexpr int SBM = 64, SBN = 64, SM_STAGES = 4;
constexpr int SM_TILE = SBM * BK;      // elements of one operand tile (8 KB)
constexpr int LDC_SM = SBN + 4;        // fp32 row stride of the epilogue's [64][68] staging image
// [k][64] image of a k-major operand: 128-byte rows, so two consecutive k share the 64 banks; the 32-byte pieces of a row are
// XOR-ed by ((k >> 1) & 1) | ((k >> 3) & 1) << 1 -- the eight k rows a tr-read half-wave touches (q and 8 + q, q = 0..3)
// then land on eight distinct 32-byte bank groups.
__device__ __forceinline__ int tr_swz64(int k) { return ((((k >> 1) & 1) | (((k >> 3) & 1) << 1))) * 16; }
template <bool TR>
__device__ __forceinline__ uint32_t small_offset1(int c, int ld, int row0, int rows) {   // chunk c of 512
  if (!TR) {   // the first 64 rows of the 128-row [row][k] image
    const int rl = c >> 3, kc = (c & 7) ^ (rl & 7);
    const int row = min(row0 + rl, rows - 1);
    return (uint32_t)(((long long)row * ld + kc * 8) * 2);
  } else {
    const int k = c >> 3, rs = (c & 7) ^ (tr_swz64(k) >> 3);
    const int row = min(row0 + rs * 8, rows - 8);
    return (uint32_t)(((long long)k * ld + row) * 2);
  }
}
template <bool TR>
__device__ __forceinline__ bf16x8 small_frag(const bf16_t* img, int row0, int kk, int lane) {
  if (!TR) {
    return load_frag<false>(img, row0, kk, lane);
  } else {
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const int k = kk * 32 + (lane >> 4) * 8 + ((lane & 15) >> 2);
    const int col = row0 + (lane & 3) * 4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + k * SBN + (col ^ tr_swz64(k))));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + (k + 4) * SBN + (col ^ tr_swz64(k + 4))));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
  }
}

template <bool TB, bool F16>
__global__ __launch_bounds__(256, 2) void gemm_small_kernel(GemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) bf16_t smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int tiles_n = (a.N + SBN - 1) / SBN, tiles_m = (a.M + SBM - 1) / SBM;
  const int nwg = tiles_n * tiles_m;
  const int orig = blockIdx.x;
  const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
  const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  const int tm = wg / tiles_n, tn = wg - tm * tiles_n;
  const int m0 = tm * SBM, n0 = tn * SBN;
  const bf16_t* __restrict__ A = a.A;
  const bf16_t* __restrict__ B = a.B;
  const int nt = a.K / BK;
  const uint32_t oA0 = small_offset1<false>(tid, a.lda, m0, a.M), oA1 = small_offset1<false>(tid + 256, a.lda, m0, a.M);
  const uint32_t oB0 = small_offset1<TB>(tid, a.ldb, n0, a.N), oB1 = small_offset1<TB>(tid + 256, a.ldb, n0, a.N);
  const long long kstepB = TB ? (long long)BK * a.ldb : BK;
  // (a stage past the end of the K range re-fetches the last tile into a buffer nobody reads again: every thread always has
  //  the same number of loads in flight, which is what the counted waits assume)
  const int lds_w = lds_addr_uniform(smem) + __builtin_amdgcn_readfirstlane(wave) * 1024;   // (scalar-side DMA addressing: see dma16_m0)
#define SM_ISSUE(T)                                                                  \
  {                                                                                  \
    const int tt = min((T), nt - 1);                                                 \
    const int dst = lds_w + ((T) % SM_STAGES) * (4 * SM_TILE);                       \
    dma16_m0(A + tt * BK, oA0, dst);                                                 \
    dma16_m0(A + tt * BK, oA1, dst + 4096);                                          \
    dma16_m0(B + tt * kstepB, oB0, dst + 2 * SM_TILE);                               \
    dma16_m0(B + tt * kstepB, oB1, dst + 2 * SM_TILE + 4096);                        \
  }
  f32x4 acc[2][2] = {};
  SM_ISSUE(0); SM_ISSUE(1); SM_ISSUE(2);
  for (int t = 0; t < nt; ++t) {
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // 4 loads per stage: stages t+1, t+2 may stay in flight
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();   // stage t has landed for every wave; every wave is done reading stage t-1
    __builtin_amdgcn_sched_barrier(0);
    SM_ISSUE(t + 3);                // -> the buffer of stage t-1
    __builtin_amdgcn_sched_barrier(0);
    const bf16_t* imgA = smem + (t % SM_STAGES) * (2 * SM_TILE);
    const bf16_t* imgB = imgA + SM_TILE;
    const bf16x8 fa00 = small_frag<false>(imgA, wr * 32, 0, lane), fa10 = small_frag<false>(imgA, wr * 32 + 16, 0, lane);
    const bf16x8 fb00 = small_frag<TB>(imgB, wc * 32, 0, lane), fb10 = small_frag<TB>(imgB, wc * 32 + 16, 0, lane);
    const bf16x8 fa01 = small_frag<false>(imgA, wr * 32, 1, lane), fa11 = small_frag<false>(imgA, wr * 32 + 16, 1, lane);
    const bf16x8 fb01 = small_frag<TB>(imgB, wc * 32, 1, lane), fb11 = small_frag<TB>(imgB, wc * 32 + 16, 1, lane);
    acc[0][0] = mfma32<F16>(fb00, fa00, acc[0][0]); acc[0][1] = mfma32<F16>(fb10, fa00, acc[0][1]);
    acc[1][0] = mfma32<F16>(fb00, fa10, acc[1][0]); acc[1][1] = mfma32<F16>(fb10, fa10, acc[1][1]);
    acc[0][0] = mfma32<F16>(fb01, fa01, acc[0][0]); acc[0][1] = mfma32<F16>(fb11, fa01, acc[0][1]);
    acc[1][0] = mfma32<F16>(fb01, fa11, acc[1][0]); acc[1][1] = mfma32<F16>(fb11, fa11, acc[1][1]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();     // no DMA in flight, no read outstanding: LDS is free for the epilogue
#undef SM_ISSUE
  float* sC = reinterpret_cast<float*>(smem);
  const int g4 = (lane >> 4) * 4, l15 = lane & 15;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
      *reinterpret_cast<f32x4*>(sC + (wr * 32 + i * 16 + l15) * LDC_SM + wc * 32 + j * 16 + g4) = acc[i][j];
  const EpiBias ebias = epi_bias(a, n0 + (tid & 7) * 8, true);
  __syncthreads();
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int chunk = tid + it * 256;
    const int rr = chunk >> 3, cc = (chunk & 7) * 8;
    const int row = m0 + rr, col = n0 + cc;
    if (row < a.M && col < a.N) epilogue_oct(a, sC + rr * LDC_SM + cc, row, col, true, 0, nullptr, ebias.b0, ebias.b1);
  }
}

// =====================================================================================================================
// Tall-tile variant of the LDS-DMA kernel: 144 x 128 tiles that ADVANCE by `mstep` rows (129...144), the rows past mstep
// being computed but not stored.  Why: the resident set is 4 workgroups x 256 CUs = 1024 tiles, and the tower-1 GEMMs
// have M = 256 molecules x 130 rows = 260 x 128 -- 1040 / 3120 / 4160 tiles, i.e. 2 / 4 / 5 rounds where 1.02 / 3.05 /
// 4.06 would do (16 tiles past a full round cost a whole extra round: 26.0 -> 36.1 us at N = K = 512).  With mstep =
// 130 the same GEMMs are exactly 1 / 3 / 4 rounds of 12.5 % taller tiles.  A row-major ([M,K]) operands only; waves
// split the tile 1 x 4 along N (144 x 32 each, 9 x 2 accumulator tiles).
constexpr int BMT = 144;
constexpr int TALL_ROWS_PER_PASS = 48;

template <bool TB, bool F16 = false>
__global__ __launch_bounds__(256, 4) void gemm_glds_tall_kernel(GemmArgs a, int mstep) {
  extern __shared__ __attribute__((aligned(16))) bf16_t smem[];
  bf16_t* imgA = smem;
  bf16_t* imgB = smem + BMT * LDT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles_n = (a.N + BN - 1) / BN, tiles_m = (a.M + mstep - 1) / mstep;
  const int nwg = tiles_n * tiles_m;
  const int orig = blockIdx.x;
  const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
  const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  const int tm = wg / tiles_n, tn = wg - tm * tiles_n;
  const int m0 = tm * mstep, n0 = tn * BN;
  const bf16_t* __restrict__ A = a.A;
  const bf16_t* __restrict__ B = a.B;
  const int ktiles = a.K / BK;
  // A tile: 144 rows x 8 chunks = 1152 chunks of 16 B -> 4.5 per thread (the fifth DMA on waves 0-1 only)
  uint32_t oA[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const int c = j * 256 + tid;
    const int rl = min(c >> 3, BMT - 1), kc = (c & 7) ^ (rl & 7);
    const int row = min(m0 + rl, a.M - 1);
    oA[j] = (uint32_t)(((long long)row * a.lda + kc * 8) * 2);
  }
  const Off4 offB = glds_offsets<TB>(a.ldb, n0, a.N, tid);
  const long long kstepB = TB ? (long long)BK * a.ldb : BK;
  f32x4 acc[9][2] = {};
#define TMF(I) acc[I][0] = mfma32<F16>(fb0, fa##I, acc[I][0]); \
               acc[I][1] = mfma32<F16>(fb1, fa##I, acc[I][1])
#define TALL_KK(KK)                                                                                   \
  {                                                                                                   \
    const bf16x8 fb0 = load_frag<TB>(imgB, wave * 32, KK, lane), fb1 = load_frag<TB>(imgB, wave * 32 + 16, KK, lane); \
    {                                                                                                 \
      const bf16x8 fa0 = load_frag<false>(imgA, 0, KK, lane), fa1 = load_frag<false>(imgA, 16, KK, lane),   \
                   fa2 = load_frag<false>(imgA, 32, KK, lane), fa3 = load_frag<false>(imgA, 48, KK, lane),  \
                   fa4 = load_frag<false>(imgA, 64, KK, lane);                                        \
      TMF(0); TMF(1); TMF(2); TMF(3); TMF(4);                                                         \
    }                                                                                                 \
    __builtin_amdgcn_sched_barrier(0);   /* two batches of A fragments: all nine at once would not fit 128 registers */ \
    {                                                                                                 \
      const bf16x8 fa5 = load_frag<false>(imgA, 80, KK, lane), fa6 = load_frag<false>(imgA, 96, KK, lane),  \
                   fa7 = load_frag<false>(imgA, 112, KK, lane), fa8 = load_frag<false>(imgA, 128, KK, lane); \
      TMF(5); TMF(6); TMF(7); TMF(8);                                                                 \
    }                                                                                                 \
  }
  // (scalar-side DMA addressing: see dma16_m0)
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  const int lds_a = lds_addr_uniform(imgA) + wave_s * 1024, lds_b = lds_addr_uniform(imgB) + wave_s * 1024;
  const char* pa = reinterpret_cast<const char*>(A);
  const char* pb = reinterpret_cast<const char*>(B);
  for (int kt = 0; kt < ktiles; ++kt) {
    dma16_m0(pa, oA[0], lds_a);
    dma16_m0(pa, oA[1], lds_a + 4096);
    dma16_m0(pa, oA[2], lds_a + 8192);
    dma16_m0(pa, oA[3], lds_a + 12288);
    if (wave_s < 2) dma16_m0(pa, oA[4], lds_a + 16384);
    glds_tile_m0(pb, offB, lds_b);
    pa += BK * 2;
    pb += kstepB * 2;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (inline-assembly DMA: not counted by the compiler)
    __syncthreads();
    TALL_KK(0);
    TALL_KK(1);
    __syncthreads();
  }
#undef TALL_KK
#undef TMF
  // epilogue: three passes of 48 rows through a [48][132] fp32 image; every wave parks its 32 columns
  const long long coff = 0;
  const int g4 = (lane >> 4) * 4, l15 = lane & 15;
  float* sC = reinterpret_cast<float*>(smem);
  float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const EpiBias ebias = epi_bias(a, n0 + (tid & 15) * 8, true);
#define TSTG(I, L) *reinterpret_cast<f32x4*>(sC + ((L) * 16 + l15) * LDC_S + wave * 32 + g4) = acc[I][0]; \
                   *reinterpret_cast<f32x4*>(sC + ((L) * 16 + l15) * LDC_S + wave * 32 + 16 + g4) = acc[I][1]
#define TPASS(P)                                                                                      \
  TSTG(3 * (P), 0); TSTG(3 * (P) + 1, 1); TSTG(3 * (P) + 2, 2);                                       \
  __syncthreads();                                                                                    \
  for (int it = 0; it < 3; ++it) {                                                                    \
    const int chunk = tid + it * 256;                                                                 \
    const int rr = chunk >> 4, cc = (chunk & 15) * 8;                                                 \
    const int rloc = (P) * TALL_ROWS_PER_PASS + rr, row = m0 + rloc, col = n0 + cc;                   \
    if (rloc < mstep && row < a.M && col < a.N)                                                       \
      epilogue_oct(a, sC + rr * LDC_S + cc, row, col, true, coff, a.colsum ? cs : nullptr, ebias.b0, ebias.b1); \
  }
  TPASS(0)
  __syncthreads();
  TPASS(1)
  __syncthreads();
  TPASS(2)
  if (a.colsum) epilogue_colsum(a, sC, cs, tid, n0);
#undef TPASS
#undef TSTG
}


// =====================================================================================================================
// 256 x 256 tile, 8 waves, LDS-DMA prefetch in flight ACROSS barriers (cdna guide 5, "the 256^2 8-phase template").
//
// The 128 x 128 kernels above run fetch -> barrier -> 16 MFMAs -> barrier with every DMA drained at each barrier and
// lean on four co-resident workgroups to overlap each other: that structure tops out near 900 TF/s, and the long-K
// weight gradients (K = 33 280 / 65 536 tokens: 35 % of the step's GEMM time) sat at 470-640 TF/s on it.  Here ONE
// workgroup owns the CU (128 KB of LDS: two K-tiles of (A | B), each 256 rows x 64 k), a K-tile is multiplied in four
// phases of 16 MFMAs per wave -- the four (A half, B half) quadrants of the wave's 128 x 64 output -- and every phase
// issues the DMA of one 16 KB piece of a LATER K-tile; the pieces are waited for with a counted s_waitcnt vmcnt(6) once
// per K-tile, so three pieces (six loads per thread) stay in flight across the raw s_barriers.
//
// Pieces.  Operand X of a K-tile is staged as two pieces X0, X1 = rows [0,128) and [128,256) of the tile, each stored as
// an ordinary 128-row tile image ([row][k] swizzled, or [k][row] for a k-major operand -- 256 contiguous bytes of global
// memory per k), so the fragment loaders of the 128 x 128 kernels read it unchanged.  Every wave works in BOTH pieces of
// each operand: wave (wr, wc) owns rows h*128 + wr*64 + [0,64) (h = 0, 1) and columns g*128 + wc*32 + [0,32) (g = 0, 1) of
// the output tile, so each phase needs only one piece of A.
// Fragments are double-buffered in registers: a phase multiplies what the PREVIOUS phase read while its own LDS reads (for
// the next phase) and its DMA piece run under the MFMAs.  A phase is (A half, one 32-deep k half) x all four column tiles
// of the wave: 4 + 4 fragments, 16 MFMAs -- 24 fragment reads per K-tile, none repeated.  K-tile t in buffer t & 1 (the
// reads of a phase are retired before its closing barrier, so a region may be restaged in the NEXT phase):
//     phase 1: MFMA A0.k0 x B.k0 | read A1.k0             | DMA A0(t+1) -> other buffer  (A0(t-1) was last read in phase 3 of t-1)
//     phase 2: MFMA A1.k0 x B.k0 | read A1.k1, B.k1       | DMA A1(t+1) -> other buffer  (A1(t-1): phase 2 of t-1)
//     phase 3: MFMA A1.k1 x B.k1 | read A0.k1             | DMA B0(t+2) -> this buffer   (B(t): phase 2); s_waitcnt vmcnt(4)
//     phase 4: MFMA A0.k1 x B.k1 | read A0.k0, B.k0 (t+1) | DMA B1(t+2) -> this buffer;                   s_waitcnt vmcnt(4)
// Phase 3's wait leaves only the two pieces issued last (A1(t+1), B0(t+2)) in flight: A0(t+1) and both B pieces of tile t+1
// have landed and the closing barrier publishes them before phase 4 reads them; phase 4's wait does the same for A1(t+1),
// read in phase 1 of tile t+1 (a staged buffer is read one phase after the wait that retires it).  Tiles past the end of
// the K range re-fetch the last tile (harmless: those regions are never read again).
constexpr int BBM = 256, BBN = 256;
constexpr int BIG_PIECE = 128 * BK;          // elements of one 16 KB piece
constexpr int BIG_BUF = 4 * BIG_PIECE;       // A0 | A1 | B0 | B1
constexpr int LDC_B = BBN + 4;               // fp32 row stride of the epilogue's [64][260] staging image

// byte offset (relative to the operand's K-tile origin) of chunk c (0..1023) of piece 0; same source-side swizzles as glds_offset1
template <bool TR>
__device__ __forceinline__ uint32_t big_offset1(int c, int ld, int row0) {
  // piece-local row r (0..127) is matrix row row0 + r (piece 1 starts 128 rows further: a wave-uniform constant)
  if (!TR) {
    const int rl = c >> 3, kc = (c & 7) ^ (rl & 7);
    return (uint32_t)(((long long)(row0 + rl) * ld + kc * 8) * 2);
  } else {
    const int k = c >> 4, rs = (c & 15) ^ (tr_swz(k) >> 3);
    return (uint32_t)(((long long)k * ld + row0 + rs * 8) * 2);
  }
}
// kbase: wave-uniform origin of the piece; d2b: byte distance of a thread's second chunk; m0v: LDS byte address of the wave's slot
__device__ __forceinline__ void big_issue(const char* kbase, long long d2b, uint32_t off, int m0v) {
  dma16_m0(kbase, off, m0v);
  dma16_m0(kbase + d2b, off, m0v + 8192);
}

// The LAST K-tile of a k-major operand whose K is not a multiple of 64 (token counts of small / ragged batches): the k rows
// past the end are fetched from a zero chunk instead -- per-lane source select; a lane's two chunks are k rows k0 and k0 + 32.
__device__ __forceinline__ void big_issue_tail(const char* kbase, long long d2b, uint32_t off, int m0v, int kv, const void* zeros, int tid) {
  const int k0 = tid >> 4;
  const char* p0 = k0 < kv ? kbase + off : reinterpret_cast<const char*>(zeros);
  const char* p1 = k0 + 32 < kv ? kbase + d2b + off : reinterpret_cast<const char*>(zeros);
  dma16_m0v(p0, m0v);
  dma16_m0v(p1, m0v + 8192);
}

// C[m][n] += sum over splits of slab[s][m][n]   (the second pass of the slab form of split-K)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ C, int M, int N, int ldc, int splits) {
  const long long n4 = (long long)M * N / 4, stride = (long long)gridDim.x * 256;
  const int n4row = N / 4;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    float4 s = *reinterpret_cast<const float4*>(slabs + i * 4);
#pragma unroll 8
    for (int k = 1; k < splits; ++k) {
      const float4 v = *reinterpret_cast<const float4*>(slabs + (long long)k * M * N + i * 4);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    const long long row = i / n4row, col = (i - row * n4row) * 4;
    float4* dst = reinterpret_cast<float4*>(C + row * ldc + col);
    float4 c = *dst;
    c.x += s.x; c.y += s.y; c.z += s.z; c.w += s.w;
    *dst = c;
  }
}

struct Frag4 {
  bf16x8 f0, f1, f2, f3;
};

// one 256 x 256 output tile (tm, tn), K-split ks of a.splitk.  F16: both operands hold fp16 (forward shapes).  BCVT: B holds fp16 and
// is converted to bf16 in registers (frag_h2bf), half behind the MFMAs of the phase that reads it from LDS, half among the MFMAs of
// the first phase that multiplies it (BIG_PHASE: POST, CV); A is bf16 (weight gradients: A = dy, B = a saved forward activation).
// TAIL (both operands k-major only): K is not a multiple of 64 -- the split that holds the last K-tile fetches it through the masked
// form; without it the loop carries no tail test at all.
template <bool TA, bool TB, bool F16 = false, bool BCVT = false, bool TAIL = false>
__device__ __forceinline__ void big_tile(const GemmArgs& a, int tm, int tn, int ks, bf16_t* smem) {
  static_assert(!(F16 && (TA || TB || BCVT)), "fp16 operands: the forward (row-major A, weight-layout B) form only");
  static_assert(!TAIL || (TA && TB), "a K tail exists with both operands k-major only");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // (scalar: DMA destinations and the bias-gradient branch stay off the vector side)
  const int wr = wave >> 2, wc = wave & 3;
  const int m0 = tm * BBM, n0 = tn * BBN;
  const int ktiles = (a.K + BK - 1) / BK;     // (a K tail only with both operands k-major: host-checked)
  const int ktail = a.K - (ktiles - 1) * BK;  // valid k rows of the last K-tile (BK: no tail)
  const int per = (ktiles + a.splitk - 1) / a.splitk;
  const int kt0 = ks * per, kt1 = min(ktiles, kt0 + per);
  if (kt0 >= kt1) return;
  const int nt = kt1 - kt0;
  const bool tail_split = TAIL && ktail != BK && kt1 == ktiles;      // this split ends on the partial tile (its local index nt - 1)
  const long long kstepA = (TA ? (long long)BK * a.lda : BK) * 2, kstepB = (TB ? (long long)BK * a.ldb : BK) * 2;     // bytes per K-tile
  const char* __restrict__ A = reinterpret_cast<const char*>(a.A) + (long long)kt0 * kstepA;
  const char* __restrict__ B = reinterpret_cast<const char*>(a.B) + (long long)kt0 * kstepB;
  // (M and N are multiples of 256 here -- host-checked -- so no source row is clamped and the eight per-thread source
  //  offsets reduce to ONE per operand plus wave-uniform constants that fold into the scalar tile base)
  const uint32_t oA = big_offset1<TA>(tid, a.lda, m0), oB = big_offset1<TB>(tid, a.ldb, n0);
  const long long dA2 = (TA ? (long long)32 * a.lda : (long long)64 * a.lda) * 2;    // second chunk of a thread (tid + 512), bytes
  const long long dB2 = (TB ? (long long)32 * a.ldb : (long long)64 * a.ldb) * 2;
  const long long hA = (TA ? 128 : (long long)128 * a.lda) * 2;                       // piece 1 vs piece 0: 128 rows further, bytes
  const long long hB = (TB ? 128 : (long long)128 * a.ldb) * 2;
  f32x4 acc[8][4] = {};
  const bool do_rs = TA && a.arowsum != nullptr && tn == 0;   // bias gradient: row sums of op(A), one row tile of each half per wave column
  const bf16x8 ones = ones_frag();
  f32x4 rs0 = {0.f, 0.f, 0.f, 0.f}, rs1 = rs0;
  // LDS layout: piece P (0..3 = A0, A1, B0, B1) of buffer Q (0, 1) at element offset (2 P + Q) * BIG_PIECE -- the two buffers of a
  // piece side by side, so that BOTH sit within the 64 KB immediate-offset reach of one fragment address register (with the buffers
  // 64 KB apart the unrolled loop needed a second set of address registers and spilled them: scratch reloads behind s_waitcnt vmcnt(0)
  // in the middle of the DMA pipeline).  lds_w: LDS byte address of this wave's 1 KB slot in piece 0 of buffer 0.
#define BIG_AT(P, Q) ((2 * (P) + (Q)) * BIG_PIECE)
  const int lds_w = lds_addr_uniform(smem) + wave * 1024;
  // K-tile origins of the pieces still to be issued: tile min(t + 1, nt - 1) of A, tile min(t + 2, nt - 1) of B (a tile index past
  // the split's end re-fetches its last tile -- harmless, those regions are never read again); advanced by scalar adds
  const char* pA1 = A + (nt > 1 ? kstepA : 0);
  const char* pB2 = B + (nt > 2 ? 2 * kstepB : (nt > 1 ? kstepB : 0));

#define BIG_WAIT(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
  // IDX: split-local index of the tile being fetched (tested only when the split ends on a partial tile)
#define BIG_ISSUE(KB, D2, OFF, IDX, M0V)                                                    \
  {                                                                                         \
    if (TAIL && tail_split && (IDX) >= nt - 1) big_issue_tail(KB, D2, OFF, M0V, ktail, a.zeros, tid); \
    else big_issue(KB, D2, OFF, M0V);                                                       \
  }
  // prologue: every piece of tile 0; of tile 1 the two B pieces (the steady state enters a tile with them in flight)
  {
    const char* pB1 = B + (nt > 1 ? kstepB : 0);
    BIG_ISSUE(A, dA2, oA, 0, lds_w + 2 * BIG_AT(0, 0));
    BIG_ISSUE(B, dB2, oB, 0, lds_w + 2 * BIG_AT(2, 0));
    BIG_ISSUE(B + hB, dB2, oB, 0, lds_w + 2 * BIG_AT(3, 0));
    BIG_ISSUE(A + hA, dA2, oA, 0, lds_w + 2 * BIG_AT(1, 0));
    BIG_ISSUE(pB1, dB2, oB, 1, lds_w + 2 * BIG_AT(2, 1));
    BIG_ISSUE(pB1 + hB, dB2, oB, 1, lds_w + 2 * BIG_AT(3, 1));
  }
  BIG_WAIT(4);
  __builtin_amdgcn_s_barrier();

  Frag4 FAx, FAy, FBx, FBy;
  // fragments of one k half (KK): A = the four 16-row tiles of the wave's rows in piece IMG -- fragment j is row tile (j + wc) & 3 of
  // the wave's 64 rows, so that the tile whose row sums this wave column takes (the bias gradient: BIG_RS) is ALWAYS f0: picking
  // "fragment wc" at run time cost a run of v_cndmask and exec-mask branches in every phase; the relabelling is free (loop-invariant
  // address arithmetic here, the same rotation where the epilogue maps accumulators to rows).  B = the wave's four 16-column tiles,
  // two from each B piece
  const int ra0 = wr * 64 + (wc & 3) * 16, ra1 = wr * 64 + ((wc + 1) & 3) * 16, ra2 = wr * 64 + ((wc + 2) & 3) * 16, ra3 = wr * 64 + ((wc + 3) & 3) * 16;
#define BIG_LDA(F, IMG, KK)                                                                                          \
  F.f0 = load_frag<TA>(IMG, ra0, KK, lane); F.f1 = load_frag<TA>(IMG, ra1, KK, lane);                                 \
  F.f2 = load_frag<TA>(IMG, ra2, KK, lane); F.f3 = load_frag<TA>(IMG, ra3, KK, lane)
#define BIG_LDB(F, Q, KK)                                                                                            \
  F.f0 = load_frag<TB>(smem + BIG_AT(2, Q), wc * 32 + 0, KK, lane); F.f1 = load_frag<TB>(smem + BIG_AT(2, Q), wc * 32 + 16, KK, lane); \
  F.f2 = load_frag<TB>(smem + BIG_AT(3, Q), wc * 32 + 0, KK, lane); F.f3 = load_frag<TB>(smem + BIG_AT(3, Q), wc * 32 + 16, KK, lane)
  // 16 MFMAs: the wave's four row tiles of A half H (acc rows H*4 + 0..3) x its four column tiles, one k half
#define BIG_MF8(H, FA, FB, R0, R1)                                                                                   \
  acc[H * 4 + R0][0] = mfma32<F16>(FB.f0, FA.f##R0, acc[H * 4 + R0][0]);                                              \
  acc[H * 4 + R1][0] = mfma32<F16>(FB.f0, FA.f##R1, acc[H * 4 + R1][0]);                                              \
  acc[H * 4 + R0][1] = mfma32<F16>(FB.f1, FA.f##R0, acc[H * 4 + R0][1]);                                              \
  acc[H * 4 + R1][1] = mfma32<F16>(FB.f1, FA.f##R1, acc[H * 4 + R1][1]);                                              \
  acc[H * 4 + R0][2] = mfma32<F16>(FB.f2, FA.f##R0, acc[H * 4 + R0][2]);                                              \
  acc[H * 4 + R1][2] = mfma32<F16>(FB.f2, FA.f##R1, acc[H * 4 + R1][2]);                                              \
  acc[H * 4 + R0][3] = mfma32<F16>(FB.f3, FA.f##R0, acc[H * 4 + R0][3]);                                              \
  acc[H * 4 + R1][3] = mfma32<F16>(FB.f3, FA.f##R1, acc[H * 4 + R1][3])
  // the same 16 MFMAs column tile by column tile (BCVT: the phases that use a B fragment set first)
#define BIG_MF4(H, FA, FB, C)                                                                                        \
  acc[H * 4 + 0][C] = mfma32<F16>(FB.f##C, FA.f0, acc[H * 4 + 0][C]);                                                 \
  acc[H * 4 + 1][C] = mfma32<F16>(FB.f##C, FA.f1, acc[H * 4 + 1][C]);                                                 \
  acc[H * 4 + 2][C] = mfma32<F16>(FB.f##C, FA.f2, acc[H * 4 + 2][C]);                                                 \
  acc[H * 4 + 3][C] = mfma32<F16>(FB.f##C, FA.f3, acc[H * 4 + 3][C])
  // one MFMA, then three of the twelve conversion instructions of the NEXT column tile's fragment, four times over
#define BIG_MIX4()                                                                                                   \
  __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);               \
  __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);               \
  __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);               \
  __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 3, 0)
  // (do_rs is a scalar condition: one s_cbranch around one MFMA)
#define BIG_RS(ACC, FA)                                                                                              \
  if (do_rs) ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, FA.f0, ACC, 0, 0, 0)
  // one phase: the LDS reads of the NEXT phase's fragments and the DMA of a later tile's piece are issued first / in
  // the middle and run under this phase's 16 MFMAs (which use fragments read one phase earlier); every read is retired
  // before the closing barrier, so the region it came from may be restaged in the next phase
  // BCVT: the four B fragments of a K half arrive from LDS as fp16 and are converted in place (frag_h2bf: 12 VALU instructions
  // each): f2, f3 behind the MFMAs of the phase that read them (POST), f0 at the top of the next phase and f1 three instructions
  // behind each of f0's MFMAs (CV, sched_group_barrier).  Every one of them shows up in the launch time wherever it is put (measured:
  // all 48 at the top of the consuming phase +14 %, behind the loading phase's MFMAs +10.6 %, interleaved +10.6 %, split +11 %; with the
  // conversion replaced by nothing +0 %, by 8 instead of 12 instructions +7 %): the loop is paced by each wave's instruction stream.
#define BIG_PHASE(H, FA, FB, RSACC, LOADS, ISSUE, WAITS, CV, POSTFB)                                                  \
  LOADS;                                                                                                             \
  __builtin_amdgcn_sched_barrier(0);                                                                                 \
  __builtin_amdgcn_s_setprio(1);                                                                                     \
  if constexpr (BCVT && CV) {                                                                                        \
    FB.f0 = frag_h2bf(FB.f0);                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                               \
    BIG_MF4(H, FA, FB, 0); FB.f1 = frag_h2bf(FB.f1); BIG_MIX4();                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                                               \
    BIG_MF4(H, FA, FB, 1);                                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                                               \
    ISSUE;                                                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                                               \
    BIG_MF4(H, FA, FB, 2);                                                                                           \
    BIG_MF4(H, FA, FB, 3);                                                                                           \
  } else {                                                                                                           \
    BIG_MF8(H, FA, FB, 0, 1);                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                               \
    ISSUE;                                                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                                               \
    BIG_MF8(H, FA, FB, 2, 3);                                                                                        \
  }                                                                                                                  \
  BIG_RS(RSACC, FA);                                                                                                 \
  __builtin_amdgcn_s_setprio(0);                                                                                     \
  if constexpr (BCVT && !(CV)) {   /* (the phases that READ a B fragment set: its second half converted under the MFMAs' drain) */ \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                                               \
    POSTFB.f2 = frag_h2bf(POSTFB.f2); POSTFB.f3 = frag_h2bf(POSTFB.f3);                                               \
    __builtin_amdgcn_sched_barrier(0);                                                                               \
  }                                                                                                                  \
  WAITS;                                                                                                             \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                 \
  __builtin_amdgcn_sched_barrier(0);                                                                                 \
  __builtin_amdgcn_s_barrier();                                                                                      \
  __builtin_amdgcn_sched_barrier(0)
  // one K-tile out of buffer Q (0 / 1; R: the other one).  Both are compile-time, so every fragment read is a base register + an
  // immediate and every DMA destination lds_w + a constant: the K loop is unrolled by two.
#define BIG_KTILE(Q, R, T)                                                                                            \
  {                                                                                                                  \
    /* c1: (A0, k 0..31)   | read A1 k-half 0            | DMA A0(t+1) -> other buffer (free since phase 4 of tile t-1) */ \
    BIG_PHASE(0, FAx, FBx, rs0, BIG_LDA(FAy, smem + BIG_AT(1, Q), 0), BIG_ISSUE(pA1, dA2, oA, (T) + 1, lds_w + 2 * BIG_AT(0, R)), (void)0, true, FBx); \
    /* c2: (A1, k 0..31)   | read A1 and B k-half 1      | DMA A1(t+1) -> other buffer (free since phase 3 of tile t-1) */ \
    BIG_PHASE(1, FAy, FBx, rs1, BIG_LDA(FAx, smem + BIG_AT(1, Q), 1); BIG_LDB(FBy, Q, 1),                             \
              BIG_ISSUE(pA1 + hA, dA2, oA, (T) + 1, lds_w + 2 * BIG_AT(1, R)), (void)0, false, FBy);                   \
    /* c3: (A1, k 32..63)  | read A0 k-half 1            | DMA B0(t+2) -> this buffer (B was last read in phase 2); wait: A0(t+1), B(t+1) landed */ \
    BIG_PHASE(1, FAx, FBy, rs1, BIG_LDA(FAy, smem + BIG_AT(0, Q), 1), BIG_ISSUE(pB2, dB2, oB, (T) + 2, lds_w + 2 * BIG_AT(2, Q)), BIG_WAIT(4), true, FBy); \
    /* c4: (A0, k 32..63)  | read A0, B of tile t+1      | DMA B1(t+2) -> this buffer; wait: A1(t+1) landed */          \
    BIG_PHASE(0, FAy, FBy, rs0, BIG_LDA(FAx, smem + BIG_AT(0, R), 0); BIG_LDB(FBx, R, 0),                             \
              BIG_ISSUE(pB2 + hB, dB2, oB, (T) + 2, lds_w + 2 * BIG_AT(3, Q)), BIG_WAIT(4), false, FBx);               \
    pA1 += ((T) + 2 < nt) ? kstepA : 0;                                                                              \
    pB2 += ((T) + 3 < nt) ? kstepB : 0;                                                                              \
  }

  BIG_LDA(FAx, smem + BIG_AT(0, 0), 0);
  BIG_LDB(FBx, 0, 0);
  if constexpr (BCVT) { FBx.f2 = frag_h2bf(FBx.f2); FBx.f3 = frag_h2bf(FBx.f3); }
  for (int t = 0; t < nt; t += 2) {
    BIG_KTILE(0, 1, t);
    if (t + 1 >= nt) break;
    BIG_KTILE(1, 0, t + 1);
  }
  BIG_WAIT(0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();   // no DMA in flight, no read outstanding: LDS is free for the epilogue
#undef BIG_KTILE
#undef BIG_AT
#undef BIG_PHASE
#undef BIG_ISSUE
#undef BIG_MF8
#undef BIG_MF4
#undef BIG_MIX4
#undef BIG_LDA
#undef BIG_LDB
#undef BIG_WAIT
  const int l15 = lane & 15, g4 = (lane >> 4) * 4;
  if (a.dbg & 1) {   // loop-only timing: keep the accumulators live, store nothing in practice
    float keep = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) keep += acc[i][0][0] + acc[i][1][1] + acc[i][2][2] + acc[i][3][3];
    if (keep == 1.2345e-30f) reinterpret_cast<float*>(a.C)[0] = keep;
    return;
  }
  if (do_rs && lane < 16) {
    // D = ones . fa^T: every register of lane m's column holds the row sum of local row m (wave column wc took row tile wc of each half)
    const int m = m0 + wr * 64 + wc * 16 + lane;
    if (m < a.M) atomicAdd(a.arowsum + m, rs0[0]);
    if (m + 128 < a.M) atomicAdd(a.arowsum + m + 128, rs1[0]);
  }
#undef BIG_RS
  const bool lead = (ks == 0);
  const long long coff = a.slab * ks;      // (slab mode: every split owns a private fp32 copy of the output)
  // accumulator (R = H*4 + i, C = G*2 + j): row m0 + H*128 + wr*64 + ((i + wc) & 3)*16 + (lane & 15) (the rotation of BIG_LDA), columns
  // n0 + G*128 + wc*32 + j*16 + 4*(lane >> 4) + {0..3}
  const int rl0 = wc & 3, rl1 = (wc + 1) & 3, rl2 = (wc + 2) & 3, rl3 = (wc + 3) & 3;
  if (a.c_dtype != MMDTI_DT_F32_ATOMIC && a.vec_ok) {
    // four passes of 64 rows through a [64][260] fp32 image; 512 threads then run the fused epilogue on 8 contiguous columns each
    float* sC = reinterpret_cast<float*>(smem);
#define BSTG(R, L, C) *reinterpret_cast<f32x4*>(sC + ((L) * 16 + l15) * LDC_B + ((C) >> 1) * 128 + wc * 32 + ((C) & 1) * 16 + g4) = acc[R][C]
#define BSTG_ROW(R, L) BSTG(R, L, 0); BSTG(R, L, 1); BSTG(R, L, 2); BSTG(R, L, 3)
#define BIG_PASS(P)                                                                                                   \
    if (wr == ((P) & 1)) { BSTG_ROW(((P) >> 1) * 4 + 0, rl0); BSTG_ROW(((P) >> 1) * 4 + 1, rl1); BSTG_ROW(((P) >> 1) * 4 + 2, rl2); BSTG_ROW(((P) >> 1) * 4 + 3, rl3); } \
    __syncthreads();                                                                                                  \
    for (int it = 0; it < 4; ++it) {                                                                                  \
      const int chunk = tid + it * 512;                                                                               \
      const int rr = chunk >> 5, cc = (chunk & 31) * 8;                                                               \
      const int row = m0 + (P) * 64 + rr, col = n0 + cc;                                                              \
      if (row < a.M && col < a.N) epilogue_oct(a, sC + rr * LDC_B + cc, row, col, lead, coff, nullptr, ebias.b0, ebias.b1); \
    }
    const EpiBias ebias = epi_bias(a, n0 + (tid & 31) * 8, lead);
    BIG_PASS(0)
    __syncthreads();
    BIG_PASS(1)
    __syncthreads();
    BIG_PASS(2)
    __syncthreads();
    BIG_PASS(3)
#undef BIG_PASS
#undef BSTG_ROW
#undef BSTG
    return;
  }
  // atomic (split-K) / unaligned outputs: a private 16 x 68 fp32 patch per wave, one row tile at a time
  float* sW = reinterpret_cast<float*>(smem) + wave * (16 * LDC_W);
#define BSTG_T(R, C) *reinterpret_cast<f32x4*>(sW + l15 * LDC_W + (C) * 16 + g4) = acc[R][C]
#define BIG_EPI(R)                                                                                   \
  {                                                                                                  \
    BSTG_T(R, 0); BSTG_T(R, 1); BSTG_T(R, 2); BSTG_T(R, 3);                                          \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                                           \
    __builtin_amdgcn_wave_barrier();                                                                 \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");                                           \
    for (int rr = 0; rr < 16; ++rr)                                                                  \
      epi_elem(a, sW[rr * LDC_W + lane], m0 + ((R) >> 2) * 128 + wr * 64 + ((((R) & 3) + wc) & 3) * 16 + rr, n0 + (lane >> 5) * 128 + wc * 32 + (lane & 31), lead, 0); \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                                           \
    __builtin_amdgcn_wave_barrier();                                                                 \
  }
  BIG_EPI(0) BIG_EPI(1) BIG_EPI(2) BIG_EPI(3) BIG_EPI(4) BIG_EPI(5) BIG_EPI(6) BIG_EPI(7)
#undef BIG_EPI
#undef BSTG_T
}

// XCD-aware bijective remap of a linear workgroup id (cdna guide T1): consecutive hardware ids go round-robin over the
// 8 XCDs, so each XCD gets a contiguous chunk of the tile list
__device__ __forceinline__ int xcd_remap(int orig, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
}

// TAIL: K % 64 != 0 (both operands k-major) -- a kernel of its own, so that the common case carries neither its tests nor its registers
template <bool TA, bool TB, bool F16 = false, bool BCVT = false, bool TAIL = false>
__global__ __launch_bounds__(512, 1) void gemm_big_kernel(GemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) bf16_t smem[];
  const int tiles_n = (a.N + BBN - 1) / BBN, tiles_m = (a.M + BBM - 1) / BBM;
  const int wg = xcd_remap(blockIdx.x, tiles_n * tiles_m);
  const int tm = wg / tiles_n;
  big_tile<TA, TB, F16, BCVT, TAIL>(a, tm, wg - tm * tiles_n, blockIdx.z, smem);
}

// ---- grouped weight gradients ------------------------------------------------------------------------------------------
// The Linears of one transformer layer produce their weight gradients dW_i[N_i, K_i] = dy_i^T . x_i over the SAME token
// range (33 280 / 65 536 rows).  One at a time, each is 4-16 output tiles of 256 x 256 and needs a 16- to 64-way K split
// to fill 256 CUs -- and every split costs a 4 MB pass over dW (fp32 partials).  Launched together the layer's GEMMs are
// 48 tiles, a 5-way split fills the chip, and the partial-sum traffic falls four-fold (255 -> 60 MB per tower-1 layer).
constexpr int GROUP_MAX = 8;
struct GroupProb {
  const bf16_t* A;      // dy  [rows, N_out] (k-major: rows = tokens)
  const bf16_t* B;      // x   [rows, N_in]
  float* C;             // dW  [N_out, ldc] fp32 (+=)
  float* arowsum;       // db  [N_out] (+=) or null
  float* slab;          // this problem's partials: [splitk][N_out][N_in] fp32
  int M, N, lda, ldb, ldc;
  int tile0, tiles_n;   // first global tile id, tiles along N
};
struct GroupArgs {
  GroupProb p[GROUP_MAX];
  int nprob, ntiles, K, splitk;
  int atomic;           // small token counts: every K-split adds its partial tile straight into dW with fp32 atomics (no slab pass)
  const void* zeros;
};

template <bool BCVT, bool TAIL>
__global__ __launch_bounds__(512, 1) void gemm_big_grouped_kernel(GroupArgs g) {
  extern __shared__ __attribute__((aligned(16))) bf16_t smem[];
  const int wg = xcd_remap(blockIdx.x, g.ntiles);
  int pi = 0;
#pragma unroll
  for (int i = 1; i < GROUP_MAX; ++i) pi += (i < g.nprob && wg >= g.p[i].tile0) ? 1 : 0;
  const GroupProb& pr = g.p[pi];
  GemmArgs a;
  a.A = pr.A; a.B = pr.B; a.C = pr.slab; a.M = pr.M; a.N = pr.N; a.K = g.K; a.lda = pr.lda; a.ldb = pr.ldb; a.ldc = pr.N;
  a.batch_inner = 1; a.sAo = a.sAi = a.sBo = a.sBi = a.sCo = a.sCi = 0;
  a.splitk = g.splitk; a.alpha = 1.f; a.beta = 0.f; a.bias = nullptr; a.residual = nullptr; a.ldr = 0; a.act = MMDTI_ACT_NONE;
  a.aux_in = nullptr; a.aux_out = nullptr; a.ld_aux = 0; a.c_dtype = MMDTI_DT_F32; a.drop_thresh = 0; a.drop_scale = 1.f; a.seed = 0; a.site = 0;
  a.vec_ok = 1; a.colsum = nullptr; a.stream_c = 0; a.dbg = 0; a.slab = (long long)pr.M * pr.N; a.arowsum = pr.arowsum; a.zeros = g.zeros;
  a.c_f16 = 0;
  if (g.atomic) { a.C = pr.C; a.ldc = pr.ldc; a.c_dtype = MMDTI_DT_F32_ATOMIC; a.slab = 0; }
  // Tile order inside a problem: the SHORT side of the tile grid runs fastest, so the ~6 consecutive tiles an XCD gets (per
  // K-split) form a compact 2 x 3 block of the output -- 5 operand pieces through that L2 instead of 7 for a 1 x 6 strip.
  const int t = wg - pr.tile0;
  const int tiles_m = (pr.M + BBM - 1) / BBM;
  int tm, tn;
  if (tiles_m < pr.tiles_n) { tn = t / tiles_m; tm = t - tn * tiles_m; }
  else { tm = t / pr.tiles_n; tn = t - tm * pr.tiles_n; }
  big_tile<true, true, false, BCVT, TAIL>(a, tm, tn, blockIdx.z, smem);
}

// The same grouped weight gradients for SMALL token counts (batches of 16-32 molecules: 1-4 k rows).  There the 256 x 256 launch is
// 48 tiles x a 5-way K split of five K-tiles each, whose partials meet in 60 MB of fp32 atomics: 62-92 us per layer, a sixth of the
// step's GPU time at 32 molecules.  Here: 64 x 64 tiles (768 of them for a tower-1 layer: every CU busy without any K split), the
// four-stage LDS-DMA ring of gemm_small_kernel with both operands k-major, the K tail zero-filled by per-lane source select, and
// dW += as a plain read-modify-write -- one workgroup owns a tile, so the result is also bitwise reproducible.
template <int STAGES, bool BCVT = false>
__global__ __launch_bounds__(256, STAGES == 3 ? 3 : 2) void gemm_small_dw_grouped_kernel(GroupArgs g) {
  extern __shared__ __attribute__((aligned(16))) bf16_t smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int wg = xcd_remap(blockIdx.x, g.ntiles);
  int pi = 0;
#pragma unroll
  for (int i = 1; i < GROUP_MAX; ++i) pi += (i < g.nprob && wg >= g.p[i].tile0) ? 1 : 0;
  const GroupProb& pr = g.p[pi];
  const int t = wg - pr.tile0;
  const int tm = t / pr.tiles_n, tn = t - tm * pr.tiles_n;
  const int m0 = tm * SBM, n0 = tn * SBN;
  const bf16_t* __restrict__ A = pr.A;     // dy [rows][N_out]
  const bf16_t* __restrict__ B = pr.B;     // x  [rows][N_in]
  const int nt = (g.K + BK - 1) / BK;
  const int ktail = g.K - (nt - 1) * BK;   // valid k rows of the last K-tile (BK: no tail)
  const uint32_t oA0 = small_offset1<true>(tid, pr.lda, m0, pr.M), oA1 = small_offset1<true>(tid + 256, pr.lda, m0, pr.M);
  const uint32_t oB0 = small_offset1<true>(tid, pr.ldb, n0, pr.N), oB1 = small_offset1<true>(tid + 256, pr.ldb, n0, pr.N);
  const long long kstepA = (long long)BK * pr.lda, kstepB = (long long)BK * pr.ldb;
  const int k0 = tid >> 3;                 // k row of this thread's first chunk; its second is k0 + 32
  // (a stage past the end of the K range re-fetches the last tile; the partial tile always through the masked form)
  const int lds_w = lds_addr_uniform(smem) + __builtin_amdgcn_readfirstlane(wave) * 1024;   // (scalar-side DMA addressing: see dma16_m0)
#define SDW_ISSUE(T)                                                                                   \
  {                                                                                                    \
    const int tt = min((T), nt - 1);                                                                   \
    const int dst = lds_w + ((T) % STAGES) * (4 * SM_TILE);                                            \
    const bf16_t* ka = A + tt * kstepA;                                                                \
    const bf16_t* kb = B + tt * kstepB;                                                                \
    if (tt == nt - 1 && ktail != BK) {                                                                 \
      const char* z = reinterpret_cast<const char*>(g.zeros);                                          \
      dma16_m0v(k0 < ktail ? reinterpret_cast<const char*>(ka) + oA0 : z, dst);                        \
      dma16_m0v(k0 + 32 < ktail ? reinterpret_cast<const char*>(ka) + oA1 : z, dst + 4096);            \
      dma16_m0v(k0 < ktail ? reinterpret_cast<const char*>(kb) + oB0 : z, dst + 2 * SM_TILE);          \
      dma16_m0v(k0 + 32 < ktail ? reinterpret_cast<const char*>(kb) + oB1 : z, dst + 2 * SM_TILE + 4096); \
    } else {                                                                                           \
      dma16_m0(ka, oA0, dst);                                                                          \
      dma16_m0(ka, oA1, dst + 4096);                                                                   \
      dma16_m0(kb, oB0, dst + 2 * SM_TILE);                                                            \
      dma16_m0(kb, oB1, dst + 2 * SM_TILE + 4096);                                                     \
    }                                                                                                  \
  }
  f32x4 acc[2][2] = {};
  const bool do_rs = pr.arowsum != nullptr && tn == 0 && wc == 0;    // bias gradient: row sums of dy^T, once per row tile
  const bf16x8 ones = ones_frag();
  f32x4 rs0 = {0.f, 0.f, 0.f, 0.f}, rs1 = rs0;
  SDW_ISSUE(0); SDW_ISSUE(1);
  if (STAGES == 4) SDW_ISSUE(2);
  for (int t2 = 0; t2 < nt; ++t2) {
    if (STAGES == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    SDW_ISSUE(t2 + STAGES - 1);
    __builtin_amdgcn_sched_barrier(0);
    const bf16_t* imgA = smem + (t2 % STAGES) * (2 * SM_TILE);
    const bf16_t* imgB = imgA + SM_TILE;
    const bf16x8 fa00 = small_frag<true>(imgA, wr * 32, 0, lane), fa10 = small_frag<true>(imgA, wr * 32 + 16, 0, lane);
    const bf16x8 fb00 = frag_cvt<BCVT>(small_frag<true>(imgB, wc * 32, 0, lane)), fb10 = frag_cvt<BCVT>(small_frag<true>(imgB, wc * 32 + 16, 0, lane));
    const bf16x8 fa01 = small_frag<true>(imgA, wr * 32, 1, lane), fa11 = small_frag<true>(imgA, wr * 32 + 16, 1, lane);
    const bf16x8 fb01 = frag_cvt<BCVT>(small_frag<true>(imgB, wc * 32, 1, lane)), fb11 = frag_cvt<BCVT>(small_frag<true>(imgB, wc * 32 + 16, 1, lane));
    acc[0][0] = mfma32<false>(fb00, fa00, acc[0][0]); acc[0][1] = mfma32<false>(fb10, fa00, acc[0][1]);
    acc[1][0] = mfma32<false>(fb00, fa10, acc[1][0]); acc[1][1] = mfma32<false>(fb10, fa10, acc[1][1]);
    acc[0][0] = mfma32<false>(fb01, fa01, acc[0][0]); acc[0][1] = mfma32<false>(fb11, fa01, acc[0][1]);
    acc[1][0] = mfma32<false>(fb01, fa11, acc[1][0]); acc[1][1] = mfma32<false>(fb11, fa11, acc[1][1]);
    if (do_rs) {
      rs0 = mfma32<false>(ones, fa00, rs0); rs1 = mfma32<false>(ones, fa10, rs1);
      rs0 = mfma32<false>(ones, fa01, rs0); rs1 = mfma32<false>(ones, fa11, rs1);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
#undef SDW_ISSUE
  if (do_rs && lane < 16) {      // (one workgroup per row tile adds: reproducible)
    atomicAdd(pr.arowsum + m0 + wr * 32 + lane, rs0[0]);
    atomicAdd(pr.arowsum + m0 + wr * 32 + 16 + lane, rs1[0]);
  }
  float* sC = reinterpret_cast<float*>(smem);
  const int g4 = (lane >> 4) * 4, l15 = lane & 15;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
      *reinterpret_cast<f32x4*>(sC + (wr * 32 + i * 16 + l15) * LDC_SM + wc * 32 + j * 16 + g4) = acc[i][j];
  __syncthreads();
#pragma unroll
  for (int it = 0; it < 4; ++it) {      // 64 rows x 16 quads: dW += tile (this workgroup owns it)
    const int chunk = tid + it * 256;
    const int rr = chunk >> 4, cc = (chunk & 15) * 4;
    float4* dst = reinterpret_cast<float4*>(pr.C + (long long)(m0 + rr) * pr.ldc + n0 + cc);
    const float4 v = *reinterpret_cast<const float4*>(sC + rr * LDC_SM + cc);
    float4 c = *dst;
    c.x += v.x; c.y += v.y; c.z += v.z; c.w += v.w;
    *dst = c;
  }
}

// dW_i += sum over splits of slab_i[s]   (all problems of a group in one launch; blockIdx.y = problem)
__global__ __launch_bounds__(256) void grouped_reduce_kernel(GroupArgs g) {
  const GroupProb& pr = g.p[blockIdx.y];
  const long long n4 = (long long)pr.M * pr.N / 4, stride = (long long)gridDim.x * 256;
  const long long slab = (long long)pr.M * pr.N;
  const int n4row = pr.N / 4;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
    for (int k = 0; k < g.splitk; ++k) {
      const float4 v = *reinterpret_cast<const float4*>(pr.slab + k * slab + i * 4);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    const long long row = i / n4row, col = (i - row * n4row) * 4;
    float4* dst = reinterpret_cast<float4*>(pr.C + row * pr.ldc + col);
    float4 c = *dst;
    c.x += s.x; c.y += s.y; c.z += s.z; c.w += s.w;
    *dst = c;
  }
}


// =====================================================================================================================
// GEMM + LayerNorm: the Linear that closes a residual branch and the LayerNorm that follows it, in ONE kernel.
//     x = residual + dropout(A.W^T + bias)          (fp32, stored: the residual stream, and the LayerNorm backward's input)
//     h = LN(x) * gamma + beta                      (bf16 for the next GEMM and / or fp32; row mean and 1/std saved)
// Pre-LN tower 1 (unicore TransformerEncoderLayer reached from models/transformers.py:137-139): out_proj -> final_layer_norm,
// fc2 -> the NEXT layer's self_attn_layer_norm (or the encoder's final_layer_norm, :160-161); post-LN BERT layers (HF RobertaLayer,
// mm_module.py:615-626): attention.output.dense -> LayerNorm, output.dense -> LayerNorm.  The separate LayerNorm kernel re-read
// the 4-byte stream the GEMM had just written (68 MB per launch at 33 280 tokens) and cost a launch per LayerNorm.
// A workgroup owns WHOLE rows (N == 512 columns: one wave per 64-column slab, 16 * R rows), so the row statistics stay on chip:
// 8 waves x (16R x 64) accumulators, the same LDS-DMA fetch -> barrier -> multiply -> barrier K loop as gemm_glds_kernel with a
// (16R | 512) x 64 tile pair (8 R + 64 KB; two workgroups per CU overlap each other).  Epilogue per 16-row slice: park the
// slice in LDS, 32 threads per row (two 8-column pieces each) form x, reduce mean and the centred second moment inside their
// half-wave (two passes, as layernorm.hip), write x, h, mean, rstd.  The dropout counters are those of the plain GEMM epilogue
// (element row * N + col), so the fused and the unfused path draw the same mask.
constexpr int LN_BN = 512;
constexpr int LDC_LN = LN_BN + 4;

struct GemmLnArgs {
  const bf16_t* A;
  const bf16_t* W;
  const float* bias;
  const float* residual;
  const float* gamma;
  const float* beta;
  float* x_out;
  float* ln_f32;
  bf16_t* ln_bf16;
  float* mean;
  float* rstd;
  int M, K, lda, ldb, ldr;
  float eps;
  uint32_t drop_thresh;
  float drop_scale;
  uint64_t seed;
  uint32_t site;
  int ln_f16;   // ln_bf16 receives fp16 values (the fp16 forward-operand mode)
};

template <int R, bool F16 = false>
__global__ __launch_bounds__(512, 4) void gemm_ln_kernel(GemmLnArgs a) {
  extern __shared__ __attribute__((aligned(16))) bf16_t smem[];
  constexpr int ROWS = 16 * R;
  bf16_t* imgA = smem;
  bf16_t* imgB = smem + ROWS * LDT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.x * ROWS;
  // A tile: ROWS x 8 chunks of 16 B (<= 640: the second DMA on the first waves only); W tile: 512 x 8 = 4096 chunks, 8 per thread
  uint32_t oA[2], oB[8];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int c = j * 512 + tid;
    const int rl = min(c >> 3, ROWS - 1), kc = (c & 7) ^ (rl & 7);
    const int row = min(m0 + rl, a.M - 1);
    oA[j] = (uint32_t)(((long long)row * a.lda + kc * 8) * 2);
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = j * 512 + tid;
    const int rl = c >> 3, kc = (c & 7) ^ (rl & 7);
    oB[j] = (uint32_t)(((long long)rl * a.ldb + kc * 8) * 2);
  }
  f32x4 acc[R][4] = {};
  const int ktiles = a.K / BK;
  constexpr int A_CHUNKS = ROWS * 8;
#define LMF(I, J) acc[I][J] = mfma32<F16>(fb##J, fa, acc[I][J])
#define LN_KK(KK)                                                                                       \
  {                                                                                                     \
    const bf16x8 fb0 = load_frag<false>(imgB, wave * 64 + 0, KK, lane), fb1 = load_frag<false>(imgB, wave * 64 + 16, KK, lane), \
                 fb2 = load_frag<false>(imgB, wave * 64 + 32, KK, lane), fb3 = load_frag<false>(imgB, wave * 64 + 48, KK, lane); \
    _Pragma("unroll") for (int I = 0; I < R; ++I) {                                                     \
      const bf16x8 fa = load_frag<false>(imgA, I * 16, KK, lane);                                       \
      LMF(I, 0); LMF(I, 1); LMF(I, 2); LMF(I, 3);                                                       \
    }                                                                                                   \
  }
  // (scalar-side DMA addressing: see dma16_m0)
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  const int lds_a = lds_addr_uniform(imgA) + wave_s * 1024, lds_b = lds_addr_uniform(imgB) + wave_s * 1024;
  const char* ab = reinterpret_cast<const char*>(a.A);
  const char* bb = reinterpret_cast<const char*>(a.W);
  for (int kt = 0; kt < ktiles; ++kt) {
    dma16_m0(ab, oA[0], lds_a);
    if (A_CHUNKS > 512 && wave_s * 64 + 512 < A_CHUNKS) dma16_m0(ab, oA[1], lds_a + 8192);
#pragma unroll
    for (int j = 0; j < 8; ++j) dma16_m0(bb, oB[j], lds_b + j * 8192);
    ab += BK * 2;
    bb += BK * 2;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (inline-assembly DMA: not counted by the compiler)
    __syncthreads();
    LN_KK(0);
    LN_KK(1);
    __syncthreads();
  }
#undef LN_KK
#undef LMF
  // ---- epilogue, 16 rows at a time through a [16][516] fp32 image (the operand tiles are dead)
  float* sC = reinterpret_cast<float*>(smem);
  const int g4 = (lane >> 4) * 4, l15 = lane & 15;
  const int er = tid >> 5, ec = (tid & 31) * 8;            // this thread's row of the slice, its two 8-column pieces: ec, ec + 256
  // bias / gamma / beta rows sit in LDS behind the staging image (48 registers otherwise, beside the live accumulators)
  float* sBias = sC + 16 * LDC_LN;
  float* sGam = sBias + LN_BN;
  float* sBet = sGam + LN_BN;
  sBias[tid] = a.bias ? a.bias[tid] : 0.f;
  sGam[tid] = a.gamma[tid];
  sBet[tid] = a.beta[tid];
#pragma unroll
  for (int I = 0; I < R; ++I) {
#pragma unroll
    for (int J = 0; J < 4; ++J) *reinterpret_cast<f32x4*>(sC + l15 * LDC_LN + wave * 64 + J * 16 + g4) = acc[I][J];
    __syncthreads();
    const int row = m0 + I * 16 + er;
    const bool ok = row < a.M;
    float v[16];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const float4 lo = *reinterpret_cast<const float4*>(sC + er * LDC_LN + ec + 256 * h);
      const float4 hi = *reinterpret_cast<const float4*>(sC + er * LDC_LN + ec + 256 * h + 4);
      float* w = v + 8 * h;
      const float4 b0 = *reinterpret_cast<const float4*>(sBias + ec + 256 * h), b1 = *reinterpret_cast<const float4*>(sBias + ec + 256 * h + 4);
      w[0] = lo.x + b0.x; w[1] = lo.y + b0.y; w[2] = lo.z + b0.z; w[3] = lo.w + b0.w;
      w[4] = hi.x + b1.x; w[5] = hi.y + b1.y; w[6] = hi.z + b1.z; w[7] = hi.w + b1.w;
      if (a.drop_thresh) {
        const uint64_t idx = (uint64_t)(ok ? row : 0) * (uint64_t)LN_BN + (uint64_t)(ec + 256 * h);
        const Rand4 r0 = philox4(a.seed, a.site, idx >> 2), r1 = philox4(a.seed, a.site, (idx >> 2) + 1);
        const uint32_t rw[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) w[e] = rw[e] >= a.drop_thresh ? w[e] * a.drop_scale : 0.f;
      }
      if (a.residual && ok) {
        const float* rp = a.residual + (long long)row * a.ldr + ec + 256 * h;
        const float4 r0 = *reinterpret_cast<const float4*>(rp), r1 = *reinterpret_cast<const float4*>(rp + 4);
        w[0] += r0.x; w[1] += r0.y; w[2] += r0.z; w[3] += r0.w; w[4] += r1.x; w[5] += r1.y; w[6] += r1.z; w[7] += r1.w;
      }
    }
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) s += v[e];
#pragma unroll
    for (int off = 1; off < 32; off <<= 1) s += __shfl_xor(s, off, 64);
    const float mean = s * (1.0f / (float)LN_BN);
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) q += (v[e] - mean) * (v[e] - mean);
#pragma unroll
    for (int off = 1; off < 32; off <<= 1) q += __shfl_xor(q, off, 64);
    const float rstd = 1.0f / sqrtf(q * (1.0f / (float)LN_BN) + a.eps);
    if (ok) {
      if ((tid & 31) == 0) {
        a.mean[row] = mean;
        a.rstd[row] = rstd;
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int col = ec + 256 * h;
        const float* w = v + 8 * h;
        float* xp = a.x_out + (long long)row * LN_BN + col;
        *reinterpret_cast<float4*>(xp) = make_float4(w[0], w[1], w[2], w[3]);
        *reinterpret_cast<float4*>(xp + 4) = make_float4(w[4], w[5], w[6], w[7]);
        float o[8];
        const float4 g0 = *reinterpret_cast<const float4*>(sGam + col), g1 = *reinterpret_cast<const float4*>(sGam + col + 4);
        const float4 e0 = *reinterpret_cast<const float4*>(sBet + col), e1 = *reinterpret_cast<const float4*>(sBet + col + 4);
        o[0] = (w[0] - mean) * rstd * g0.x + e0.x; o[1] = (w[1] - mean) * rstd * g0.y + e0.y;
        o[2] = (w[2] - mean) * rstd * g0.z + e0.z; o[3] = (w[3] - mean) * rstd * g0.w + e0.w;
        o[4] = (w[4] - mean) * rstd * g1.x + e1.x; o[5] = (w[5] - mean) * rstd * g1.y + e1.y;
        o[6] = (w[6] - mean) * rstd * g1.z + e1.z; o[7] = (w[7] - mean) * rstd * g1.w + e1.w;
        if (a.ln_f32) {
          float* lp = a.ln_f32 + (long long)row * LN_BN + col;
          *reinterpret_cast<float4*>(lp) = make_float4(o[0], o[1], o[2], o[3]);
          *reinterpret_cast<float4*>(lp + 4) = make_float4(o[4], o[5], o[6], o[7]);
        }
        if (a.ln_bf16) {
          uint4 u;
          u.x = pack16x2(a.ln_f16, o[0], o[1]);
          u.y = pack16x2(a.ln_f16, o[2], o[3]);
          u.z = pack16x2(a.ln_f16, o[4], o[5]);
          u.w = pack16x2(a.ln_f16, o[6], o[7]);
          *reinterpret_cast<uint4*>(a.ln_bf16 + (long long)row * LN_BN + col) = u;
        }
      }
    }
    __syncthreads();   // the next slice overwrites the image
  }
}

}  // namespace mmdti

MMDTI_DEFINE_SALT_PULL(gemm)
using namespace mmdti;

static int g_gemm_big = getenv("MMDTI_GEMM_BIG") ? atoi(getenv("MMDTI_GEMM_BIG")) : 1;
static int g_gemm_small = getenv("MMDTI_GEMM_SMALL") ? atoi(getenv("MMDTI_GEMM_SMALL")) : 1;   // 64 x 64 tiles for small launches
static int g_gemm_deep = getenv("MMDTI_GEMM_DEEP") ? atoi(getenv("MMDTI_GEMM_DEEP")) : 1;      // four-stage ring at <= 1 workgroup per CU

// 256 zero bytes in device memory (see GemmArgs::zeros); allocated at the first call that needs it
static const void* zero_page() {
  static void* page = nullptr;
  if (!page) {
    if (hipMalloc(&page, 256) != hipSuccess) { page = nullptr; return nullptr; }
    if (hipMemset(page, 0, 256) != hipSuccess) { (void)hipFree(page); page = nullptr; return nullptr; }
  }
  return page;
}
static int g_gemm_dbg = 0;     // measurement only: 1 = gemm_big_kernel returns after its K loop (no epilogue, no slab pass)

extern "C" int mmdti_set_option(const char* name, int value) {
  MMDTI_REQUIRE(name != nullptr, "set_option: null name");
  if (strcmp(name, "gemm_big") == 0) { g_gemm_big = value; return MMDTI_OK; }
  if (strcmp(name, "gemm_dbg") == 0) { g_gemm_dbg = value; return MMDTI_OK; }
  if (strcmp(name, "gemm_small") == 0) { g_gemm_small = value; return MMDTI_OK; }
  if (strcmp(name, "gemm_deep") == 0) { g_gemm_deep = value; return MMDTI_OK; }
  set_error("set_option: unknown option '%s'", name);
  return MMDTI_ERR_INVALID;
}

// Shapes on which the 256 x 256 kernel (one workgroup per CU) beats the 128 x 128 ones (four per CU): enough tiles to fill
// the 256 CUs with little waste in the last round (measured table: DESIGN.md section 4 "GEMM").
static bool big_shape_pays(int M, int N, int K, int splitk, int transA, int transB, bool reads_aux) {
  // Measured on MI355X against the 128 x 128 kernels (scratch/gemm_big_test.py, profiles/r02_gemm_big_ab.json).  The K loop
  // of this kernel runs at 900-1300 TF/s, but with ONE workgroup per CU nothing overlaps a tile's epilogue (an HBM / VALU
  // burst of 15-20 us for a 256 x 256 fp32 / GELU tile) with another tile's loop, and 33 280-row outputs quantise badly on
  // 256 CUs (130 row tiles).  It pays where the loop dominates and the tile count divides the chip:
  const long long tiles = (long long)cdiv(M, 256) * cdiv(N, 256);
  if (splitk > 1) return K >= 16384 && 256 % tiles == 0 && tiles >= 4;     // long-K weight gradients: 4 or 16 output tiles (x1.02...1.25)
  // tower-2 shapes in whole rounds: forward x1.03...1.10; input gradients (k-major B) x1.09...1.10 when the epilogue reads nothing
  // (with the saved-activation multiply the 128 x 128 kernels win, 266 vs 304 us); 512 x 512 x 512 stays with them too (72 vs 77 us)
  return tiles % 256 == 0 && K >= 512 && N <= 2048 && M >= 65536 && (long long)N * K >= 512 * 1024 && !(transB && reads_aux);
}

extern "C" int mmdti_gemm_bf16(mmdti_stream_t stream, const void* A, const void* B, void* C, int M, int N, int K,
                               int lda, int ldb, int ldc, int transA, int transB, int batch_outer, int batch_inner,
                               long long sAo, long long sAi, long long sBo, long long sBi, long long sCo,
                               long long sCi, int splitk, float alpha, float beta, const float* bias,
                               const float* residual, int ldr, int act, const void* aux_in, void* aux_out,
                               int ld_aux, int c_dtype, float drop_p, unsigned long long seed, unsigned int site,
                               float* colsum_out, float* arowsum_out, void* workspace, long long workspace_bytes) {
  MMDTI_REQUIRE(M > 0 && N > 0 && K > 0, "gemm: M,N,K must be positive (got %d,%d,%d)", M, N, K);
  MMDTI_REQUIRE(A && B && C, "gemm: null operand");
  MMDTI_REQUIRE(aligned16(A) && aligned16(B), "gemm: A and B must be 16-byte aligned");
  MMDTI_REQUIRE(lda % 8 == 0 && ldb % 8 == 0, "gemm: lda/ldb must be multiples of 8 elements (got %d,%d)", lda, ldb);
  MMDTI_REQUIRE(sAo % 8 == 0 && sAi % 8 == 0 && sBo % 8 == 0 && sBi % 8 == 0, "gemm: batch strides must be multiples of 8");
  MMDTI_REQUIRE(batch_outer >= 1 && batch_inner >= 1 && splitk >= 1, "gemm: batch/splitk must be >= 1");
  const bool ab16 = (c_dtype & MMDTI_DT_AB_F16) != 0;       // A and B hold fp16 (forward shapes only)
  const bool bcvt = (c_dtype & MMDTI_DT_B_F16) != 0;        // B holds fp16, converted to bf16 in registers (weight-gradient shapes only)
  c_dtype &= ~(MMDTI_DT_AB_F16 | MMDTI_DT_B_F16);
  const int c_f16 = c_dtype == MMDTI_DT_F16;
  if (c_f16) c_dtype = MMDTI_DT_BF16;
  MMDTI_REQUIRE(c_dtype == MMDTI_DT_F32 || c_dtype == MMDTI_DT_BF16 || c_dtype == MMDTI_DT_F32_ATOMIC, "gemm: bad c_dtype");
  MMDTI_REQUIRE(!ab16 || (!transA && !transB && splitk == 1 && !arowsum_out && batch_outer * batch_inner == 1),
                "gemm: fp16 operands are built for the forward Linear shapes (row-major A, weight-layout B, no split-K, no batch)");
  MMDTI_REQUIRE(!bcvt || (!ab16 && transA && transB && batch_outer * batch_inner == 1 && c_dtype == MMDTI_DT_F32_ATOMIC),
                "gemm: an fp16 B beside a bf16 A is built for the weight-gradient shapes (both operands k-major, fp32 accumulation into C, no batch)");
  MMDTI_REQUIRE(!c_f16 || !colsum_out, "gemm: colsum_out with an fp16 output is not supported");
  MMDTI_REQUIRE(splitk == 1 || c_dtype == MMDTI_DT_F32_ATOMIC, "gemm: splitk>1 needs the atomic fp32 output mode");
  MMDTI_REQUIRE(splitk == 1 || (act == MMDTI_ACT_NONE && drop_p == 0.f), "gemm: splitk>1 cannot fuse act/dropout");
  MMDTI_REQUIRE((act != MMDTI_ACT_GELU_BWD && act != MMDTI_ACT_MUL_AUX) || aux_in, "gemm: gelu_bwd / mul_aux need aux_in");
  MMDTI_REQUIRE(act != MMDTI_ACT_GELU_G || aux_out, "gemm: gelu_g needs aux_out");
  MMDTI_REQUIRE(act == MMDTI_ACT_NONE || act == MMDTI_ACT_GELU || act == MMDTI_ACT_GELU_BWD || act == MMDTI_ACT_GELU_G || act == MMDTI_ACT_MUL_AUX,
                "gemm: unknown act %d", act);
  MMDTI_REQUIRE(batch_outer * batch_inner == 1 || (!residual && !aux_in && !aux_out && drop_p == 0.f),
                "gemm: residual/aux/dropout epilogues are unbatched only");
  MMDTI_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "gemm: dropout p out of range");
  GemmArgs a;
  a.A = (const bf16_t*)A; a.B = (const bf16_t*)B; a.C = C;
  a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc;
  a.batch_inner = batch_inner;
  a.sAo = sAo; a.sAi = sAi; a.sBo = sBo; a.sBi = sBi; a.sCo = sCo; a.sCi = sCi;
  a.splitk = splitk; a.alpha = alpha; a.beta = beta; a.bias = bias; a.residual = residual; a.ldr = ldr;
  a.act = act; a.aux_in = (const bf16_t*)aux_in; a.aux_out = (bf16_t*)aux_out; a.ld_aux = ld_aux; a.c_dtype = c_dtype; a.c_f16 = c_f16;
  a.drop_thresh = dropout_thresh(drop_p); a.drop_scale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  a.seed = seed; a.site = site;
  a.colsum = colsum_out;
  a.arowsum = nullptr;
  a.zeros = nullptr;
  a.slab = 0;
  a.dbg = g_gemm_dbg;
  {
    // streaming stores for outputs of at least MMDTI_GEMM_STREAM_MB (default 96 MB; 0 = always, negative = never)
    static const long long stream_mb = getenv("MMDTI_GEMM_STREAM_MB") ? atoll(getenv("MMDTI_GEMM_STREAM_MB")) : 96;
    const long long cbytes = (long long)M * N * (c_dtype == MMDTI_DT_BF16 ? 2 : 4) * batch_outer * batch_inner;
    a.stream_c = (stream_mb >= 0 && cbytes >= stream_mb * 1000000LL && beta == 0.f) ? 1 : 0;
  }
  MMDTI_REQUIRE(!arowsum_out || (transA && batch_outer * batch_inner == 1), "gemm: arowsum_out needs a k-major A (transA) and no batch");
  {
    const bool bf = c_dtype == MMDTI_DT_BF16;
    const int cal = bf ? 8 : 4;
    bool ok = aligned16(C) && (ldc % cal == 0) && (sCo % cal == 0) && (sCi % cal == 0) && (N % 8 == 0);
    if (bias) ok = ok && aligned16(bias);
    if (residual) ok = ok && aligned16(residual) && (ldr % 4 == 0);
    if (aux_in) ok = ok && aligned16(aux_in) && (ld_aux % 8 == 0);
    if (aux_out) ok = ok && aligned16(aux_out) && (ld_aux % 8 == 0);
    a.vec_ok = ok ? 1 : 0;
  }
  MMDTI_REQUIRE(!colsum_out || (a.vec_ok && splitk == 1 && batch_outer * batch_inner == 1 && c_dtype != MMDTI_DT_F32_ATOMIC),
                "gemm: colsum_out needs the aligned, unbatched, unsplit output path");
  const int tiles = cdiv(M, BM) * cdiv(N, BN);
  dim3 grid(tiles, 1, batch_outer * batch_inner * splitk), block(256);
  MMDTI_REQUIRE(grid.z <= 65535u, "gemm: batch*splitk too large (%u)", grid.z);
  // two (A|B) tile buffers, re-used by the epilogue's per-wave patches
  // one (A|B) tile pair (32,768 B), re-used by the epilogue's [64][132] fp32 staging image (33,792 B)
  const size_t smem = (size_t)64 * LDC_S * sizeof(float);
  hipStream_t s = (hipStream_t)stream;
  // > 64 KiB of dynamic LDS: opt in once per instantiation.
  typedef void (*kern_t)(GemmArgs);
  static const kern_t kerns[2][2][2] = {
      {{gemm_bf16_kernel<false, false, false>, gemm_bf16_kernel<false, false, true>},
       {gemm_bf16_kernel<false, true, false>, gemm_bf16_kernel<false, true, true>}},
      {{gemm_bf16_kernel<true, false, false>, gemm_bf16_kernel<true, false, true>},
       {gemm_bf16_kernel<true, true, false>, gemm_bf16_kernel<true, true, true>}}};
  static bool attr_done = false;
  if (!attr_done) {
    for (int i = 0; i < 8; ++i) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kerns[i >> 2][(i >> 1) & 1][i & 1]),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) {
        set_error("gemm: hipFuncSetAttribute(MaxDynamicSharedMemorySize=%zu) failed", smem);
        return MMDTI_ERR_LAUNCH;
      }
    }
    attr_done = true;
  }
  // bare-load fast path: no K tail, no ragged 8-row chunk on a k-major operand, offsets fit 32 bits
  const bool fast = (K % BK == 0) && (!transA || M % 8 == 0) && (!transB || N % 8 == 0) && M >= 8 && N >= 8 &&
                    ((long long)(transA ? BK : M) * lda * 2 < 0x7fffffffLL) && ((long long)(transB ? BK : N) * ldb * 2 < 0x7fffffffLL);
  static const kern_t gkerns[2][2] = {{gemm_glds_kernel<false, false, false>, gemm_glds_kernel<false, true, false>},
                                      {gemm_glds_kernel<true, false, false>, gemm_glds_kernel<true, true, false>}};
  static const kern_t gkerns2[2][2] = {{gemm_glds_kernel<false, false, true>, gemm_glds_kernel<false, true, true>},
                                       {gemm_glds_kernel<true, false, true>, gemm_glds_kernel<true, true, true>}};
  static const kern_t gkerns3[2][2] = {{gemm_glds_kernel<false, false, 2>, gemm_glds_kernel<false, true, 2>},
                                       {gemm_glds_kernel<true, false, 2>, gemm_glds_kernel<true, true, 2>}};
  // LDS-DMA tile fetch for every bare-load shape except the split-K weight gradients (measured: -15...-20 % on the
  // N >= 1536 / K >= 1536 shapes, equal at 512x512, +9 % on the atomic split-K ones); MMDTI_GEMM_GLDS=0 turns it off
  static const int use_glds = getenv("MMDTI_GEMM_GLDS") ? atoi(getenv("MMDTI_GEMM_GLDS")) : 1;
  // split-K weight gradients: double-buffered DMA from 48 output tiles up (-5...-13 %), register staging below (+13 %)
  // arowsum rides on the kernel that has register room for it (double-buffered DMA: the large weight gradients); on
  // the other paths it is the plain column-sum pass over A's memory image ([K][M] row-major)
  // (a small split-K weight gradient that also carries its bias gradient takes the double-buffered kernel too: +6 us
  //  there against a 35-50 us column-sum pass over dy)
  const bool dbuf_path = fast && use_glds && ((splitk > 1 && (tiles >= 48 || (arowsum_out && transA))) || use_glds == 3);
  // 256 x 256 tiles with the DMA in flight across barriers (gemm_big_kernel): MMDTI_GEMM_BIG=0 off, 1 (default) where
  // the shape fills the chip, 2 every eligible shape
  const int use_big = g_gemm_big;
  const bool big_ok = fast && use_big && batch_outer * batch_inner == 1 && !colsum_out && M >= 256 && N >= 256 &&
                      (c_dtype == MMDTI_DT_F32_ATOMIC || a.vec_ok) && M % 256 == 0 && N % 256 == 0;
  if (big_ok && (use_big == 2 || big_shape_pays(M, N, K, splitk, transA, transB, aux_in != nullptr))) {
    typedef void (*bkern_t)(GemmArgs);
    static const bkern_t bkerns[2][2] = {{gemm_big_kernel<false, false>, gemm_big_kernel<false, true>},
                                         {gemm_big_kernel<true, false>, gemm_big_kernel<true, true>}};
    // [4]: fp16 operands (forward shapes); [5]: fp16 B converted in registers (weight gradients)
    static const bkern_t bkerns_all[6] = {bkerns[0][0], bkerns[0][1], bkerns[1][0], bkerns[1][1], gemm_big_kernel<false, false, true>,
                                          gemm_big_kernel<true, true, false, true>};
    const bkern_t bk = ab16 ? bkerns_all[4] : (bcvt ? bkerns_all[5] : bkerns[transA ? 1 : 0][transB ? 1 : 0]);
    const size_t smem_b = (size_t)2 * BIG_BUF * sizeof(bf16_t);
    static bool big_attr = false;
    if (!big_attr) {
      for (int i = 0; i < 6; ++i)
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(bkerns_all[i]), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_b) != hipSuccess) {
          set_error("gemm: hipFuncSetAttribute(MaxDynamicSharedMemorySize=%zu) failed", smem_b);
          return MMDTI_ERR_LAUNCH;
        }
      big_attr = true;
    }
    const int btiles = cdiv(M, BBM) * cdiv(N, BBN);
    int sk = splitk;
    if (splitk > 1) {   // weight gradients: about one workgroup per CU, at least 4 K-tiles per split
      sk = max(1, min(K / BK / 4, 256 / btiles));      // floor: one round of workgroups (a 257th would cost a whole second round)
      const int kts = K / BK, per = cdiv(kts, sk);
      sk = cdiv(kts, per);          // no empty split (the slab form sums EVERY slab)
    }
    a.splitk = sk;
    a.arowsum = arowsum_out;
    dim3 bgrid(btiles, 1, sk);
    const long long slab = (long long)M * N;
    const bool slabs = sk > 1 && workspace && aligned16(workspace) && workspace_bytes >= (long long)sk * slab * 4 && N % 8 == 0 &&
                       alpha == 1.f && !bias && !residual && act == MMDTI_ACT_NONE && slab % 4 == 0 && aligned16(C) && ldc % 4 == 0;
    if (slabs && !(g_gemm_dbg & 1)) {
      // split ks stores its partial tile into slab ks (vector epilogue, plain stores); splitk_reduce_kernel adds the sum into C
      GemmArgs p = a;
      p.C = workspace; p.ldc = N; p.c_dtype = MMDTI_DT_F32; p.beta = 0.f; p.vec_ok = 1; p.stream_c = 0; p.slab = slab;
      hipLaunchKernelGGL(bk, bgrid, dim3(512), smem_b, s, p);
      const long long n4 = slab / 4;
      hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)min((n4 + 255) / 256, 4096LL)), dim3(256), 0, s, (const float*)workspace,
                         reinterpret_cast<float*>(C), M, N, ldc, sk);
    } else {
      hipLaunchKernelGGL(bk, bgrid, dim3(512), smem_b, s, a);
    }
    MMDTI_LAUNCH_CHECK();
    return MMDTI_OK;
  }
  if (arowsum_out) {
    if (dbuf_path || (bcvt && fast && use_glds)) {
      a.arowsum = arowsum_out;
    } else if (int e = mmdti_colsum_bf16(stream, A, K, M, lda, arowsum_out)) {
      return e;
    }
  }
  if (bcvt) {
    // weight gradient with an fp16 activation operand: the double-buffered LDS-DMA kernel on bare-load shapes (whatever the split), the
    // register-staged one otherwise
    if (fast && use_glds) {
      a.arowsum = arowsum_out;
      hipLaunchKernelGGL((gemm_glds_kernel<true, true, 1, false, true>), grid, block, 4 * (size_t)BM * LDT * sizeof(bf16_t), s, a);
    } else {
      static bool attr_cv = false;
      if (!attr_cv) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_kernel<true, true, false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_kernel<true, true, true, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) {
          set_error("gemm: hipFuncSetAttribute(MaxDynamicSharedMemorySize=%zu) failed", smem);
          return MMDTI_ERR_LAUNCH;
        }
        attr_cv = true;
      }
      if (fast) hipLaunchKernelGGL((gemm_bf16_kernel<true, true, true, false, true>), grid, block, smem, s, a);
      else hipLaunchKernelGGL((gemm_bf16_kernel<true, true, false, false, true>), grid, block, smem, s, a);
    }
  }
  else if (dbuf_path)
    hipLaunchKernelGGL(gkerns2[transA ? 1 : 0][transB ? 1 : 0], grid, block, 4 * (size_t)BM * LDT * sizeof(bf16_t), s, a);
  else if (fast && use_glds && splitk == 1) {
    // tall tiles when they save a whole round of the 1024 resident workgroups (see gemm_glds_tall_kernel)
    int mstep = 0;
    static const int use_tall = getenv("MMDTI_GEMM_TALL") ? atoi(getenv("MMDTI_GEMM_TALL")) : 1;
    const int use_small = g_gemm_small;
    static const int small_max_tiles = getenv("MMDTI_GEMM_SMALL_TILES") ? atoi(getenv("MMDTI_GEMM_SMALL_TILES")) : 128;
    const int use_deep = g_gemm_deep;
    const int deep_max_wgs = 256;                     // (one workgroup per CU)
    if (use_tall && !transA && a.vec_ok && grid.z == 1 && M >= 1024) {
      const int slots = 1024, tn = cdiv(N, BN);
      const int r128 = cdiv(tiles, slots);
      if (r128 >= 2 && (r128 - 1) * slots >= tn) {
        const int rows_fit = ((r128 - 1) * slots) / tn;           // row-tiles that fit in one round fewer
        const int sneed = cdiv(M, rows_fit);
        // (measured: 2 -> 1 rounds is -12...-16 %; 4 -> 3 and 5 -> 4 rounds lose to the taller tile's own cost)
        if (sneed > 128 && sneed <= BMT && 1.125 * (r128 - 1) < 0.75 * r128) mstep = sneed;
      }
    }
    if (mstep) {
      grid.x = cdiv(M, mstep) * cdiv(N, BN);
      const size_t smem_t = (size_t)(BMT + BN) * LDT * sizeof(bf16_t);
      if (ab16) hipLaunchKernelGGL((gemm_glds_tall_kernel<false, true>), grid, block, smem_t, s, a, mstep);
      else if (transB) hipLaunchKernelGGL(gemm_glds_tall_kernel<true>, grid, block, smem_t, s, a, mstep);
      else hipLaunchKernelGGL(gemm_glds_tall_kernel<false>, grid, block, smem_t, s, a, mstep);
    } else if (use_small && !transA && a.vec_ok && c_dtype != MMDTI_DT_F32_ATOMIC && !colsum_out && grid.z == 1 && tiles <= small_max_tiles &&
               !(ab16 && transB)) {
      // small launches: a quarter of the tile per workgroup, four times the CUs (see gemm_small_kernel)
      const dim3 sgrid(cdiv(M, SBM) * cdiv(N, SBN));
      const size_t smem_s = (size_t)SM_STAGES * 2 * SM_TILE * sizeof(bf16_t);
      if (ab16) hipLaunchKernelGGL((gemm_small_kernel<false, true>), sgrid, block, smem_s, s, a);
      else if (transB) hipLaunchKernelGGL((gemm_small_kernel<true, false>), sgrid, block, smem_s, s, a);
      else hipLaunchKernelGGL((gemm_small_kernel<false, false>), sgrid, block, smem_s, s, a);
    } else if (use_deep && tiles * (int)grid.z <= deep_max_wgs && K >= 4 * BK) {
      // at most one workgroup per CU: the four-stage ring hides the fetch latency nothing else would (small batches)
      const size_t smem_d = (size_t)DEEP_STAGES * 2 * BM * LDT * sizeof(bf16_t);
      static bool deep_attr = false;
      if (!deep_attr) {
        const void* fns[5] = {reinterpret_cast<const void*>(gkerns3[0][0]), reinterpret_cast<const void*>(gkerns3[0][1]),
                              reinterpret_cast<const void*>(gkerns3[1][0]), reinterpret_cast<const void*>(gkerns3[1][1]),
                              reinterpret_cast<const void*>(gemm_glds_kernel<false, false, 2, true>)};
        for (int i = 0; i < 5; ++i)
          if (hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_d) != hipSuccess) {
            set_error("gemm: hipFuncSetAttribute(MaxDynamicSharedMemorySize=%zu) failed", smem_d);
            return MMDTI_ERR_LAUNCH;
          }
        deep_attr = true;
      }
      if (ab16) hipLaunchKernelGGL((gemm_glds_kernel<false, false, 2, true>), grid, block, smem_d, s, a);
      else hipLaunchKernelGGL(gkerns3[transA ? 1 : 0][transB ? 1 : 0], grid, block, smem_d, s, a);
    } else if (ab16) {
      hipLaunchKernelGGL((gemm_glds_kernel<false, false, false, true>), grid, block, smem, s, a);
    } else {
      hipLaunchKernelGGL(gkerns[transA ? 1 : 0][transB ? 1 : 0], grid, block, smem, s, a);
    }
  }
  else if (ab16) {
    static bool attr16 = false;
    if (!attr16) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_kernel<false, false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess ||
          hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_kernel<false, false, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) {
        set_error("gemm: hipFuncSetAttribute(MaxDynamicSharedMemorySize=%zu) failed", smem);
        return MMDTI_ERR_LAUNCH;
      }
      attr16 = true;
    }
    if (fast) hipLaunchKernelGGL((gemm_bf16_kernel<false, false, true, true>), grid, block, smem, s, a);
    else hipLaunchKernelGGL((gemm_bf16_kernel<false, false, false, true>), grid, block, smem, s, a);
  }
  else
    hipLaunchKernelGGL(kerns[transA ? 1 : 0][transB ? 1 : 0][fast ? 1 : 0], grid, block, smem, s, a);
  MMDTI_LAUNCH_CHECK();
  return MMDTI_OK;
}


extern "C" int mmdti_gemm_ln_bf16(mmdti_stream_t stream, const void* A_bf16, const void* W_bf16, const float* bias, const float* residual,
                                  int M, int N, int K, int lda, int ldb, int ldr, float drop_p, unsigned long long seed, unsigned int site,
                                  float* x_out, const float* gamma, const float* beta, float eps, float* ln_f32, void* ln_bf16,
                                  float* mean, float* rstd, int f16) {
  MMDTI_REQUIRE(A_bf16 && W_bf16 && x_out && gamma && beta && mean && rstd && (ln_f32 || ln_bf16), "gemm_ln: null argument");
  MMDTI_REQUIRE(M > 0 && N == LN_BN && K > 0 && K % BK == 0, "gemm_ln: N must be %d and K a multiple of %d (got M=%d N=%d K=%d)", LN_BN, BK, M, N, K);
  MMDTI_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && lda >= K && ldb >= K && aligned16(A_bf16) && aligned16(W_bf16), "gemm_ln: operand strides / alignment");
  MMDTI_REQUIRE(aligned16(x_out) && aligned16(gamma) && aligned16(beta) && (!bias || aligned16(bias)) && (!ln_f32 || aligned16(ln_f32)) &&
                    (!ln_bf16 || aligned16(ln_bf16)) && (!residual || (aligned16(residual) && ldr % 4 == 0 && ldr >= N)),
                "gemm_ln: 16-byte alignment required");
  MMDTI_REQUIRE((long long)M * lda * 2 < 0x7fffffffLL && (long long)N * ldb * 2 < 0x7fffffffLL, "gemm_ln: operand too large for 32-bit offsets");
  MMDTI_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "gemm_ln: dropout p out of range");
  GemmLnArgs a;
  a.A = (const bf16_t*)A_bf16; a.W = (const bf16_t*)W_bf16; a.bias = bias; a.residual = residual; a.gamma = gamma; a.beta = beta;
  a.x_out = x_out; a.ln_f32 = ln_f32; a.ln_bf16 = (bf16_t*)ln_bf16; a.mean = mean; a.rstd = rstd;
  a.M = M; a.K = K; a.lda = lda; a.ldb = ldb; a.ldr = ldr; a.eps = eps;
  a.drop_thresh = dropout_thresh(drop_p); a.drop_scale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  a.seed = seed; a.site = site;
  a.ln_f16 = (f16 >> 1) & 1;          // bit 0: A and W are fp16; bit 1: the 16-bit LayerNorm output is fp16
  const bool ab16 = f16 & 1;
  // rows per tile: 64 or 80 -- whichever needs less (rounds of the 512 resident workgroups) x (rows per tile)
  static const int force_r = getenv("MMDTI_GEMM_LN_ROWS") ? atoi(getenv("MMDTI_GEMM_LN_ROWS")) : 0;
  const long long cost4 = (long long)cdiv(cdiv(M, 64), 512) * 4, cost5 = (long long)cdiv(cdiv(M, 80), 512) * 5;
  const int R = force_r == 64 ? 4 : (force_r == 80 ? 5 : (cost5 < cost4 ? 5 : 4));
  const size_t smem = max((size_t)(16 * R + LN_BN) * LDT * sizeof(bf16_t), (size_t)(16 * LDC_LN + 3 * LN_BN) * sizeof(float));
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_ln_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_ln_kernel<5>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_ln_kernel<4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_ln_kernel<5, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) != hipSuccess) {
      set_error("gemm_ln: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");
      return MMDTI_ERR_LAUNCH;
    }
    attr = true;
  }
  if (ab16) {
    if (R == 5) hipLaunchKernelGGL((gemm_ln_kernel<5, true>), dim3(cdiv(M, 80)), dim3(512), smem, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((gemm_ln_kernel<4, true>), dim3(cdiv(M, 64)), dim3(512), smem, (hipStream_t)stream, a);
  } else if (R == 5) hipLaunchKernelGGL(gemm_ln_kernel<5>, dim3(cdiv(M, 80)), dim3(512), smem, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(gemm_ln_kernel<4>, dim3(cdiv(M, 64)), dim3(512), smem, (hipStream_t)stream, a);
  MMDTI_LAUNCH_CHECK();
  return MMDTI_OK;
}

extern "C" int mmdti_linear_dw_grouped(mmdti_stream_t stream, int nprob, const void* const* dy_bf16, const void* const* x_bf16,
                                       float* const* dw, float* const* db, const int* n_out, const int* n_in, const int* ldy,
                                       const int* ldx, const int* lddw, int rows, void* workspace, long long workspace_bytes, int x_f16) {
  MMDTI_REQUIRE(nprob >= 1 && nprob <= GROUP_MAX, "linear_dw_grouped: 1..%d problems (got %d)", GROUP_MAX, nprob);
  MMDTI_REQUIRE(dy_bf16 && x_bf16 && dw && n_out && n_in && ldy && ldx && lddw, "linear_dw_grouped: null argument table");
  MMDTI_REQUIRE(rows >= 64, "linear_dw_grouped: at least 64 rows (got %d)", rows);   // (any count: a K tail is zero-filled in the kernel)
  MMDTI_REQUIRE(workspace && aligned16(workspace), "linear_dw_grouped: a 16-byte aligned workspace is required");
  GroupArgs g;
  g.nprob = nprob; g.K = rows;
  g.zeros = zero_page();
  MMDTI_REQUIRE(g.zeros != nullptr, "linear_dw_grouped: could not allocate the zero page");
  int tiles = 0;
  long long elems = 0;
  for (int i = 0; i < nprob; ++i) {
    MMDTI_REQUIRE(dy_bf16[i] && x_bf16[i] && dw[i], "linear_dw_grouped: null operand in problem %d", i);
    MMDTI_REQUIRE(n_out[i] > 0 && n_in[i] > 0 && n_out[i] % BBM == 0 && n_in[i] % BBN == 0, "linear_dw_grouped: dimensions must be multiples of 256 (problem %d: %d x %d)", i, n_out[i], n_in[i]);
    MMDTI_REQUIRE(ldy[i] % 8 == 0 && ldx[i] % 8 == 0 && lddw[i] % 4 == 0 && aligned16(dy_bf16[i]) && aligned16(x_bf16[i]) && aligned16(dw[i]),
                  "linear_dw_grouped: alignment (problem %d)", i);
    MMDTI_REQUIRE((long long)BK * ldy[i] * 2 < 0x7fffffffLL && (long long)BK * ldx[i] * 2 < 0x7fffffffLL, "linear_dw_grouped: row stride too large");
    GroupProb& p = g.p[i];
    p.A = (const bf16_t*)dy_bf16[i]; p.B = (const bf16_t*)x_bf16[i]; p.C = dw[i]; p.arowsum = db ? db[i] : nullptr;
    p.M = n_out[i]; p.N = n_in[i]; p.lda = ldy[i]; p.ldb = ldx[i]; p.ldc = lddw[i];
    p.tile0 = tiles; p.tiles_n = n_in[i] / BBN;
    tiles += (n_out[i] / BBM) * p.tiles_n;
    elems += (long long)n_out[i] * n_in[i];
  }
  for (int i = nprob; i < GROUP_MAX; ++i) { g.p[i] = g.p[0]; g.p[i].tile0 = 0x7fffffff; }
  hipStream_t s = (hipStream_t)stream;
  // small token counts: 64 x 64 tiles, no K split, plain += (gemm_small_dw_grouped_kernel)
  static const int small_rows = getenv("MMDTI_GROUPED_SMALL_ROWS") ? atoi(getenv("MMDTI_GROUPED_SMALL_ROWS")) : 4096;
  if (rows <= small_rows && g_gemm_small) {
    GroupArgs gs = g;
    int st = 0;
    for (int i = 0; i < nprob; ++i) {
      gs.p[i].tile0 = st; gs.p[i].tiles_n = n_in[i] / SBN;
      st += (n_out[i] / SBM) * gs.p[i].tiles_n;
    }
    gs.ntiles = st; gs.splitk = 1; gs.atomic = 0;
    // (three stages = 48 KB: three workgroups per CU; measured -2 % on the step against a four-stage ring at two per CU)
    if (x_f16) hipLaunchKernelGGL((gemm_small_dw_grouped_kernel<3, true>), dim3(st), dim3(256), (size_t)3 * 2 * SM_TILE * sizeof(bf16_t), s, gs);
    else hipLaunchKernelGGL((gemm_small_dw_grouped_kernel<3, false>), dim3(st), dim3(256), (size_t)3 * 2 * SM_TILE * sizeof(bf16_t), s, gs);
    MMDTI_LAUNCH_CHECK();
    return MMDTI_OK;
  }
  g.ntiles = tiles;
  const int kts = cdiv(rows, BK);
  int sk = max(1, min(kts / 4, 256 / max(1, tiles)));   // floor: all workgroups resident in ONE round
  sk = cdiv(kts, cdiv(kts, sk));                     // no empty split: every slab is summed
  g.splitk = sk;
  MMDTI_REQUIRE(workspace_bytes >= (long long)sk * elems * 4, "linear_dw_grouped: workspace too small (%lld bytes needed for %d splits)",
                (long long)sk * elems * 4, sk);
  float* ws = reinterpret_cast<float*>(workspace);
  for (int i = 0; i < nprob; ++i) {
    g.p[i].slab = ws;
    ws += (long long)sk * g.p[i].M * g.p[i].N;
  }
  const size_t smem_b = (size_t)2 * BIG_BUF * sizeof(bf16_t);
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_big_grouped_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_b) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_big_grouped_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_b) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_big_grouped_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_b) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_big_grouped_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_b) != hipSuccess) {
      set_error("linear_dw_grouped: hipFuncSetAttribute(MaxDynamicSharedMemorySize=%zu) failed", smem_b);
      return MMDTI_ERR_LAUNCH;
    }
    attr = true;
  }
  // Small token counts (the reference's real batch sizes, 16-32 molecules): the step is a chain of ~20 us kernels, and the slab
  // pass is one more of them per layer -- the K-splits add into dW with fp32 atomics instead (a few MB of them: cheaper than a launch)
  // (reached only with the 64 x 64 kernel switched off: gemm_small = 0)
  g.atomic = rows <= 4096 ? 1 : 0;
  const bool ktail = rows % BK != 0;
  if (x_f16) {
    if (ktail) hipLaunchKernelGGL((gemm_big_grouped_kernel<true, true>), dim3(tiles, 1, sk), dim3(512), smem_b, s, g);
    else hipLaunchKernelGGL((gemm_big_grouped_kernel<true, false>), dim3(tiles, 1, sk), dim3(512), smem_b, s, g);
  } else {
    if (ktail) hipLaunchKernelGGL((gemm_big_grouped_kernel<false, true>), dim3(tiles, 1, sk), dim3(512), smem_b, s, g);
    else hipLaunchKernelGGL((gemm_big_grouped_kernel<false, false>), dim3(tiles, 1, sk), dim3(512), smem_b, s, g);
  }
  if (!g.atomic) {
    long long max_n4 = 0;
    for (int i = 0; i < nprob; ++i) max_n4 = max(max_n4, (long long)g.p[i].M * g.p[i].N / 4);
    hipLaunchKernelGGL(grouped_reduce_kernel, dim3((unsigned)min((max_n4 + 255) / 256, 2048LL), nprob), dim3(256), 0, s, g);
  }
  MMDTI_LAUNCH_CHECK();
  return MMDTI_OK;
}

/* splits the grouped weight-gradient launch will use for `tiles` output tiles over `rows` tokens (workspace sizing) */
extern "C" int mmdti_linear_dw_grouped_splits(int tiles, int rows) {
  if (tiles <= 0 || rows < BK) return 1;
  const int kts = cdiv(rows, BK);
  const int sk = max(1, min(kts / 4, 256 / max(1, tiles)));
  return cdiv(kts, cdiv(kts, sk));
}
