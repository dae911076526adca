import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/mm-dti_amd')
from mmdti_hip import ops
def run(B, N, H, lens, p):
    D, ld, scale = H * 8, ops.pair_ld(N), 8 ** -0.5
    nt = ops.pair_tiles(N)
    g = torch.Generator().manual_seed(1)
    qkv = torch.randn(B, N, 3 * D, generator=g).bfloat16().cuda().view(B * N, 3 * D)
    dO = torch.randn(B, N, D, generator=g).bfloat16().cuda().view(B * N, D)
    key_pad = torch.zeros(B, N, dtype=torch.bool)
    for b, n in enumerate(lens): key_pad[b, n:] = True
    bias = torch.zeros(B, H, N, ld); bias[..., :N] = torch.randn(B, H, N, N, generator=g)
    bias_t = ops.pair_tile(bias.cuda(), N, float("-inf")).half()
    kt = torch.tensor([(n + 15) // 16 for n in lens], dtype=torch.int32, device="cuda")
    kw = dict(drop_p=p, seed=5, site=3)
    s_d, o_d = ops.pair_attn_fwd(qkv, bias_t, key_pad.cuda(), B, N, H, ld, scale, **kw)
    s_r, o_r = ops.pair_attn_fwd(qkv, bias_t, key_pad.cuda(), B, N, H, ld, scale, key_tiles=kt, rag_store=True, **kw)
    g_in = torch.zeros(B, H, N, ld); g_in[..., :N] = torch.randn(B, H, N, N, generator=g).masked_fill(key_pad.view(B, 1, 1, N), 0.0)
    g_d = ops.pair_tile(g_in.cuda(), N, 0.0); g_r = g_d.clone()
    dq_d = ops.pair_attn_bwd(qkv, s_d, dO, g_d, B, N, H, ld, scale, False, **kw).view(B, N, 3, H, 8).float()
    s_p = s_r.clone()
    for b in range(B):
        s_p[b, :, :, ops.pair_key_tiles_effective(int(kt[b]), nt):] = float("nan")
    dq_r = ops.pair_attn_bwd(qkv, s_p, dO, g_r, B, N, H, ld, scale, False, key_tiles=kt, **kw).view(B, N, 3, H, 8).float()
    ne = (dq_d != dq_r) & ~(dq_d.isnan() & dq_r.isnan())
    print('   nan in ragged dq:', int(dq_r.isnan().sum()))
    print(f"B={B} N={N} H={H} lens={lens} p={p}: fwd o equal {torch.equal(o_d, o_r)}; dqkv mismatches {int(ne.sum())}")
    if ne.any():
        idx = ne.nonzero()
        for col, name in enumerate(["b", "row", "q|k|v", "head", "d"]):
            u = torch.unique(idx[:, col]).tolist()
            print("   ", name, u[:20], "..." if len(u) > 20 else "")
        i = idx[0]; print("   first", i.tolist(), dq_d[tuple(i)].item(), dq_r[tuple(i)].item(), " max abs diff", float((dq_d - dq_r).abs().max()))
        print("    G equal:", torch.equal(g_d, g_r))
for p in (0.0, 0.1):
    run(3, 100, 64, (100, 17, 81), p)
    run(3, 100, 8, (100, 17, 81), p)
run(4, 130, 8, (130, 37, 64, 5), 0.1)
