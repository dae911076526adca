import sys, torch
sys.path.insert(0, "mm-dti_amd")
from mmdti_hip import ops
B, N, H, lens = 4, 130, 8, (130, 37, 64, 5)
D, ld, scale = H * 8, ops.pair_ld(N), 8 ** -0.5
nt = ops.pair_tiles(N)
g = torch.Generator().manual_seed(1)
qkv = torch.randn(B, N, 3 * D, generator=g).bfloat16().cuda().view(B * N, 3 * D)
key_pad = torch.zeros(B, N, dtype=torch.bool)
for b, n in enumerate(lens): key_pad[b, n:] = True
bias = torch.zeros(B, H, N, ld); bias[..., :N] = torch.randn(B, H, N, N, generator=g)
bias_t = ops.pair_tile(bias.cuda(), N, float("-inf"))
kt = torch.tensor([(n + 15) // 16 for n in lens], dtype=torch.int32, device="cuda")
s_d, o_d = ops.pair_attn_fwd(qkv, bias_t, key_pad.cuda(), B, N, H, ld, scale)
s_r, o_r = ops.pair_attn_fwd(qkv, bias_t, key_pad.cuda(), B, N, H, ld, scale, key_tiles=kt, rag_store=True)
ne = ~((s_d == s_r) | (s_d.isnan() & s_r.isnan()))
v = ne.view(B, H, nt, nt, 4, 16, 4)
idx = v.nonzero()
print("mismatches", idx.shape[0])
for col, name in enumerate(["b", "h", "tq", "tk", "kq", "q", "k4"]):
    print(name, torch.unique(idx[:, col]).tolist())
i = idx[0]; print(i.tolist(), s_d.view(B, H, nt, nt, 4, 16, 4)[tuple(i)].item(), s_r.view(B, H, nt, nt, 4, 16, 4)[tuple(i)].item())
qq = idx[:, 2] * 16 + idx[:, 5]; kk = idx[:, 3] * 16 + idx[:, 4] * 4 + idx[:, 6]
inside = (qq < N) & (kk < N)
print("inside NxN:", int(inside.sum()))
ii = idx[inside]
if ii.shape[0]:
    for col, name in enumerate(["b", "h", "tq", "tk", "kq", "q", "k4"]):
        print(name, torch.unique(ii[:, col]).tolist())
    for b in range(B):
        sel = ii[ii[:, 0] == b]
        print("b", b, "kt", int(kt[b]), "tk", torch.unique(sel[:, 3]).tolist(), "n", sel.shape[0])
    i = ii[0]; print(i.tolist(), s_d.view(B, H, nt, nt, 4, 16, 4)[tuple(i)].item(), s_r.view(B, H, nt, nt, 4, 16, 4)[tuple(i)].item())
print("o equal", torch.equal(o_d, o_r))
