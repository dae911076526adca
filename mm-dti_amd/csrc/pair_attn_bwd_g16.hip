// pair_attn_bwd_g16.hip -- the pair-attention backward's hot path with the gradient chain G stored as bf16 (layout 7): the same kernel and
// launcher as pair_attn_bwd.hip (pair_attn_bwd_mfma.h), compiled as its own translation unit so that the two builds run in parallel.
#include "pair_attn_bwd_mfma.h"

MMDTI_DEFINE_SALT_PULL(pair_attn_bwd_g16)

namespace mmdti {
void pa_bwd_compact_launch_g16(int nqb, dim3 grid, dim3 blk, hipStream_t st, const void* qkv_bf16, const void* s, const void* do_bf16, void* g,
                               void* dqkv_bf16, int N, int H, int ld, float scale, int g_in_zero, uint32_t th8, float sc,
                               unsigned long long seed, unsigned int site, const int* key_tiles, const int* row_off, int qkv_f16) {
  pa_bwd_compact_launch<__bf16>(nqb, grid, blk, st, qkv_bf16, s, do_bf16, g, dqkv_bf16, N, H, ld, scale, g_in_zero, th8, sc, seed, site, key_tiles, row_off, qkv_f16);
}
}  // namespace mmdti
