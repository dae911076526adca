"""Cost of the ragged kernels' per-tile branches: RAG kernel with key_tiles == all tiles (nothing skipped) vs the dense kernel."""
import os, sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/mm-dti_amd')
from mmdti_hip import ops
B, H = 256, 64
def t(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n
for N in (64, 96, 113, 130):
    D = H * 8; ld = ops.pair_ld(N); scale = 8 ** -0.5; nt = ops.pair_tiles(N)
    qkv = torch.randn(B * N, 3 * D, device='cuda').bfloat16()
    bias = ops.pair_tile(torch.randn(B, H, N, ld, device='cuda'), N).half()
    do = torch.randn(B * N, D, device='cuda').bfloat16()
    g = torch.zeros(bias.shape, device='cuda')
    full = torch.full((B,), nt, dtype=torch.int32, device='cuda')
    half = torch.full((B,), max(1, nt // 2), dtype=torch.int32, device='cuda')
    s_out, o = ops.pair_attn_fwd(qkv, bias, None, B, N, H, ld, scale, 0.1, 1, 1)
    row = [N, nt]
    for kt in (None, full, half):
        row.append(t(lambda: ops.pair_attn_fwd(qkv, bias, None, B, N, H, ld, scale, 0.1, 1, 1, key_tiles=kt)))
        row.append(t(lambda: ops.pair_attn_bwd(qkv, s_out, do, g, B, N, H, ld, scale, False, 0.1, 1, 1, key_tiles=kt)))
    print("N=%d nt=%d | dense fwd %.3f bwd %.3f | RAG all tiles fwd %.3f bwd %.3f | RAG half the tiles fwd %.3f bwd %.3f" % tuple(row))
