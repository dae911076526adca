import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/mm-dti_amd')
from mmdti_hip import ops
G = lambda s: torch.Generator().manual_seed(s)
dev = lambda t: t.cuda()
bf = lambda t: t.to(torch.bfloat16)
def run(B, N, H, lens, p, rep):
    D, ld, scale = H * 8, ops.pair_ld(N), 8 ** -0.5
    nt = ops.pair_tiles(N)
    qkv = dev(bf(torch.randn(B, N, 3 * D, generator=G(1)))).view(B * N, 3 * D)
    dO = dev(bf(torch.randn(B, N, D, generator=G(3)))).view(B * N, D)
    key_pad = torch.zeros(B, N, dtype=torch.bool)
    for b, n in enumerate(lens): key_pad[b, n:] = True
    bias = torch.zeros(B, H, N, ld); bias[..., :N] = torch.randn(B, H, N, N, generator=G(2))
    bias_t = ops.pair_tile(dev(bias), N, float("-inf")).half()
    kt = torch.tensor([(n + 15) // 16 for n in lens], dtype=torch.int32, device="cuda")
    ke = [ops.pair_key_tiles_effective(int(k), nt) for k in kt]
    kw = dict(drop_p=p, seed=5, site=3)
    s_d, o_d = ops.pair_attn_fwd(qkv, bias_t, dev(key_pad), B, N, H, ld, scale, **kw)
    s_n, o_n = ops.pair_attn_fwd(qkv, bias_t, dev(key_pad), B, N, H, ld, scale, key_tiles=kt, rag_store=False, **kw)
    s_poison = s_n.clone()
    for b in range(B): s_poison[b, :, :, ke[b]:] = float("nan")
    g_in = torch.zeros(B, H, N, ld); g_in[..., :N] = torch.randn(B, H, N, N, generator=G(4)).masked_fill(key_pad.view(B, 1, 1, N), 0.0)
    g_0 = ops.pair_tile(dev(g_in), N, 0.0)
    for it in range(rep):
        g_d = g_0.clone(); g_r = g_0.clone()
        dq_d = ops.pair_attn_bwd(qkv, s_d, dO, g_d, B, N, H, ld, scale, False, **kw).view(B, N, 3, H, 8).float()
        dq_r = ops.pair_attn_bwd(qkv, s_poison, dO, g_r, B, N, H, ld, scale, False, key_tiles=kt, **kw).view(B, N, 3, H, 8).float()
        dq_d2 = ops.pair_attn_bwd(qkv, s_d, dO, g_0.clone(), B, N, H, ld, scale, False, **kw).view(B, N, 3, H, 8).float()
        ne = (dq_d != dq_r)
        print(f"it {it}: ragged-vs-dense mismatches {int(ne.sum())}  nan {int(dq_r.isnan().sum())}   dense-vs-dense {int((dq_d != dq_d2).sum())}")
        if ne.any():
            idx = ne.nonzero()
            for col, name in enumerate(["b", "row", "q|k|v", "head", "d"]):
                u = torch.unique(idx[:, col]).tolist()
                print("   ", name, u[:24], "..." if len(u) > 24 else "")
            i = idx[0]; print("   first", i.tolist(), dq_d[tuple(i)].item(), dq_r[tuple(i)].item(), " max abs diff", float((dq_d - dq_r).abs().max()))
run(3, 100, 64, (100, 17, 81), 0.1, 3)
