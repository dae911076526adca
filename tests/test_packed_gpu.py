"""GPU tests of the PACKED token layout (mmdti_hip/packing.py): ragged batches run on every sequence's real tokens plus ONE
representative pad row, instead of every padded row of the reference's right-padded tensors (models/mm_model.py:645-682,
models/transformers.py:114-118, models/infonce.py:32-33).

Kernel level: each packed entry point against its dense form on the same data -- bit for bit where the arithmetic is the same
(pair attention, fused attention), to summation-order tolerance where it is not (weighted mean, pool).
Model level: at dropout 0 the packed step equals the padded step (losses to 1e-6 relative, gradients inside the run-to-run band
of the atomics); the reference-fixture tests (test_g9_gpu.py / G10) run through both layouts.
"""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import mmdti_oracle as O
from g9util import refarch_cfg, product_model, load_fixture_weights, rel_l2, cosine
from test_kernels_gpu import ops, dev, bf, rt, close, G          # noqa: F401  (fixture + helpers)


def _pack(lens, S):
    from mmdti_hip.packing import PackedRows
    return PackedRows(torch.tensor(lens), S, device="cuda")


# ------------------------------------------------------------------------------------------------ host layout on the device
def test_packed_rows_pack_unpack_roundtrip():
    pk = _pack([5, 9, 1, 9], 9)
    assert pk.M == 6 + 9 + 2 + 9 and pk.max_rows == 9 and pk.off.tolist() == [0, 6, 15, 17, 26] and pk.n_real.tolist() == [5, 9, 1, 9]
    x = torch.arange(4 * 9, device="cuda").view(4, 9)
    xp = pk.pack(x)
    assert xp.tolist() == [0, 1, 2, 3, 4, 5] + list(range(9, 18)) + [18, 19] + list(range(27, 36))
    back = pk.unpack(xp)
    assert back[0].tolist() == [0, 1, 2, 3, 4, 5, 5, 5, 5] and back[2].tolist() == [18] + [19] * 8 and torch.equal(back[1], x[1])
    assert pk.row_seq.tolist() == [0] * 6 + [1] * 9 + [2] * 2 + [3] * 9
    assert pk.pad_weights().tolist() == [1] * 5 + [4] + [1] * 9 + [1, 8] + [1] * 9


# ------------------------------------------------------------------------------------------------ pair attention
@pytest.mark.parametrize("B,N,H,lens,p", [(4, 130, 8, (130, 37, 64, 5), 0.0), (3, 100, 64, (100, 17, 81), 0.1), (2, 258, 8, (40, 258), 0.0),
                                         (3, 33, 8, (16, 32, 15), 0.0), (2, 16, 8, (3, 16), 0.1), (5, 150, 16, (1, 150, 149, 96, 48), 0.0)])
def test_pair_attn_packed_rows_equal_padded_rows(ops, B, N, H, lens, p):
    """Packed token rows (row_off) against the ragged kernels on the padded rows, same data: the real rows and the representative
    pad row (padded position n_b) of O, dq, dk, dv and their S / G rows are equal bit for bit; the pad rows past the
    representative one are not written (S / G of those query rows keep what they held)."""
    D, ld, scale = H * 8, ops.pair_ld(N), 8 ** -0.5
    nt = ops.pair_tiles(N)
    pk = _pack(lens, N)
    x = torch.randn(B, N, 3 * D, generator=G(1))
    dO = torch.randn(B, N, D, generator=G(3))
    key_pad = torch.zeros(B, N, dtype=torch.bool)
    for b, n in enumerate(lens):
        key_pad[b, n:] = True
        x[b, n:] = x[b, n:n + 1] if n < N else x[b, n:]            # the reference's pad rows: identical within a molecule
        dO[b, n:] = dO[b, n:n + 1] if n < N else dO[b, n:]
    qkv, dO = dev(bf(x)).view(B * N, 3 * D), dev(bf(dO)).view(B * N, D)
    bias = torch.zeros(B, H, N, ld); bias[..., :N] = torch.randn(B, H, N, N, generator=G(2))
    for b, n in enumerate(lens):
        bias[b, :, n:, :N] = bias[b, :, n:n + 1, :N] if n < N else bias[b, :, n:, :N]
    bias_t = ops.pair_tile(dev(bias), N, float("-inf")).half()
    kt = torch.tensor([(n + 15) // 16 for n in lens], dtype=torch.int32, device="cuda")
    ke = [ops.pair_key_tiles_effective(int(k), nt) for k in kt]
    kw = dict(drop_p=p, seed=5, site=3)
    gat = pk.gather
    qkv_p, dO_p, pad_p = qkv[gat].contiguous(), dO[gat].contiguous(), dev(key_pad).view(-1)[gat].contiguous()

    def rowsof(t):                 # tiled plane -> [B, H, N, 16 * nt] rows (slots past N are never compared)
        u = torch.zeros(B, H, N, nt * 16, device=t.device)
        u[..., :N] = ops.pair_untile(t, N).float()
        return u

    s_d, o_d = ops.pair_attn_fwd(qkv, bias_t, dev(key_pad), B, N, H, ld, scale, key_tiles=kt, **kw)
    s_p, o_p = ops.pair_attn_fwd(qkv_p, bias_t, pad_p, B, N, H, ld, scale, key_tiles=kt, row_off=pk.off, **kw)
    assert o_p.shape == (pk.M, D)
    assert torch.equal(o_p, o_d[gat])           # (dropout counters are positional: the representative row IS padded position n_b, mask included)
    for b, n in enumerate(lens):
        r = min(n + 1, N)
        assert torch.equal(rowsof(s_p)[b, :, :r, :16 * ke[b]], rowsof(s_d)[b, :, :r, :16 * ke[b]])
    # second layer on the packed layer's own S, with poison in everything the packed kernels must not read
    s_poison = s_p.clone()
    for b, n in enumerate(lens):
        r = min(n + 1, N)
        q_blocks = (r + 15) // 16
        s_poison[b][:, ops.pair_slots(N, "cuda", q_lo=16 * q_blocks)] = float("nan")       # query blocks past the representative row
        s_poison[b][:, ops.pair_slots(N, "cuda", k_lo=16 * ke[b])] = float("nan")          # key tiles past the effective count
    s2_d, o2_d = ops.pair_attn_fwd(qkv, s_d, None, B, N, H, ld, scale, key_tiles=kt, **kw)
    s2_p, o2_p = ops.pair_attn_fwd(qkv_p, s_poison, None, B, N, H, ld, scale, key_tiles=kt, row_off=pk.off, **kw)
    assert torch.equal(o2_p, o2_d[gat]) and bool(torch.isfinite(o2_p.float()).all())
    # backward: G chain zero-initialised (ragged contract); the padded run's pad rows all receive the representative row's dO
    g_in = torch.zeros(B, H, N, ld); g_in[..., :N] = torch.randn(B, H, N, N, generator=G(4)).masked_fill(key_pad.view(B, 1, 1, N), 0.0)
    for b, n in enumerate(lens):
        g_in[b, :, n:, :N] = g_in[b, :, n:n + 1, :N] if n < N else g_in[b, :, n:, :N]
    g_0 = ops.pair_tile(dev(g_in), N, 0.0)
    g_d, g_p = g_0.clone(), g_0.clone()
    dq_d = ops.pair_attn_bwd(qkv, s_d, dO, g_d, B, N, H, ld, scale, False, key_tiles=kt, **kw)
    dq_p = ops.pair_attn_bwd(qkv_p, s_poison, dO_p, g_p, B, N, H, ld, scale, False, key_tiles=kt, row_off=pk.off, **kw)
    assert dq_p.shape == (pk.M, 3 * D) and bool(torch.isfinite(dq_p.float()).all())
    # dq of every packed row = the padded run's; dk / dv: the padded run sums the contributions of ALL its pad query rows, the
    # packed run has one of them -- so compare them where there is no padding, and the query part everywhere
    assert torch.equal(dq_p[:, :D], dq_d[gat][:, :D])
    for b, n in enumerate(lens):
        r = min(n + 1, N)
        assert torch.equal(rowsof(g_p)[b, :, :r, :16 * ke[b]], rowsof(g_d)[b, :, :r, :16 * ke[b]])
        if n < N:       # the representative pad row is no key: its dk / dv rows are written, as zeros
            row = int(pk.off[b]) + n
            assert float(dq_p[row, D:].float().abs().max()) == 0.0
    # dk / dv exactly: a padded run whose surplus pad rows carry dO = 0 and G_in = 0 (one pad row left) is the packed run
    dO1 = dO.view(B, N, D).clone()
    g1 = g_in.clone()
    for b, n in enumerate(lens):
        dO1[b, n + 1:] = 0
        g1[b, :, n + 1:] = 0
    g1_t = ops.pair_tile(dev(g1), N, 0.0)
    dq_1 = ops.pair_attn_bwd(qkv, s_d, dO1.view(B * N, D).contiguous(), g1_t, B, N, H, ld, scale, False, key_tiles=kt, **kw)
    # (the surplus pad rows of that padded run contribute P * (dP - delta) + G_in with dP = delta = G_in = 0: exactly zero, with or
    #  without dropout)
    assert torch.equal(dq_p, dq_1[gat])


def test_pair_attn_packed_rejects_bad_arguments(ops):
    B, N, H = 2, 20, 8
    pk = _pack([20, 7], N)
    qkv = torch.zeros(pk.M, 3 * H * 8, device="cuda", dtype=torch.bfloat16)
    bias = torch.zeros(B, H, ops.pair_plane(N), device="cuda", dtype=torch.float16)
    kt = torch.tensor([2, 1], dtype=torch.int32, device="cuda")
    with pytest.raises(ops.MMDTIError):
        ops.pair_attn_fwd(qkv, bias, None, B, N, H, 20, 0.35, row_off=pk.off)                        # row_off without key_tiles
    with pytest.raises(ops.MMDTIError):
        ops.pair_attn_fwd(qkv, bias, torch.zeros(B * N, device="cuda", dtype=torch.bool), B, N, H, 20, 0.35, key_tiles=kt, row_off=pk.off)
    with pytest.raises(ops.MMDTIError):
        ops.pair_attn_fwd(qkv[:-1], bias, None, B, N, H, 20, 0.35, key_tiles=kt)                     # dense layout with the wrong row count


# ------------------------------------------------------------------------------------------------ fused attention
@pytest.mark.parametrize("B,heads,hd,qlens,klens,Sq,Sk,p", [(3, 4, 64, (37, 5, 20), (37, 5, 20), 37, 37, 0.0), (4, 16, 32, (130, 12, 77, 1), (50, 256, 3, 100), 130, 256, 0.0),
                                                          (2, 8, 64, (256, 100), (256, 100), 256, 256, 0.1), (2, 2, 32, (9, 16), (160, 1), 16, 160, 0.2)])
def test_attn_fused_packed_sequences_equal_masked_dense(ops, B, heads, hd, qlens, klens, Sq, Sk, p):
    """Packed sequences (q_off / k_off / k_cnt) against the dense kernels with an additive finfo.min key mask on the padded rows:
    every packed row's context, dq and the real rows' dk / dv are equal bit for bit (the dense run's masked keys have probability
    exactly 0); the representative pad row of the key side receives dk = dv = 0."""
    D = heads * hd
    scale = 1.0 / math.sqrt(hd)
    pq, pkk = _pack(qlens, Sq), _pack(klens, Sk)
    self_attn = qlens == klens and Sq == Sk
    q = rt(torch.randn(B, Sq, D, generator=G(1)) * 1.5)
    k = rt(torch.randn(B, Sk, D, generator=G(2)) * 1.5)
    v = rt(torch.randn(B, Sk, D, generator=G(3)) * 1.5)
    do = rt(torch.randn(B, Sq, D, generator=G(4)))
    for b, n in enumerate(qlens):          # surplus padded query rows get no upstream gradient (the dense run then matches one pad row)
        do[b, n + 1:] = 0
    mask = torch.zeros(B, Sk)
    for b, n in enumerate(klens):
        mask[b, :n] = 1
    add = (1 - mask) * torch.finfo(torch.float32).min
    flat = lambda t, L: dev(bf(t.reshape(B * L, D)))
    qd, kd, vd, dod = flat(q, Sq), flat(k, Sk), flat(v, Sk), flat(do, Sq)
    vl = ops.AttnVarlen(pq, pkk)
    kw = dict(drop_p=p, seed=77, site=2)
    # NOTE: the dropout quad index is (query position) * (key tiles of the kernel instantiation * 4) + ...: both runs must take
    # the same instantiation -- they do when the longest real key count and the padded key length fall in the same class
    ctx_d, st_d = ops.attn_fwd(qd, kd, vd, dev(add), B, heads, Sq, Sk, scale, **kw)
    ctx_p, st_p = ops.attn_fwd(qd[pq.gather].contiguous(), kd[pkk.gather].contiguous(), vd[pkk.gather].contiguous(), None, B, heads,
                               vl.Lq, vl.Lk, scale, vl=vl, **kw)
    same_class = (vl.Lk <= 160) == (Sk <= 160)
    if p == 0.0 or same_class:
        assert torch.equal(ctx_p, ctx_d[pq.gather])
    dq_d, dk_d, dv_d = ops.attn_bwd(qd, kd, vd, dev(add), dod, st_d, B, heads, Sq, Sk, scale, **kw)
    dq_p, dk_p, dv_p = ops.attn_bwd(qd[pq.gather].contiguous(), kd[pkk.gather].contiguous(), vd[pkk.gather].contiguous(), None,
                                    dod[pq.gather].contiguous(), st_p, B, heads, vl.Lq, vl.Lk, scale, vl=vl, **kw)
    assert bool(torch.isfinite(dk_p.float()).all()) and bool(torch.isfinite(dv_p.float()).all())
    if p == 0.0 or same_class:
        assert torch.equal(dq_p, dq_d[pq.gather])
        assert torch.equal(dk_p, dk_d[pkk.gather]) and torch.equal(dv_p, dv_d[pkk.gather])
    for b, n in enumerate(klens):
        if n < Sk:
            row = int(pkk.off[b]) + n
            assert float(dk_p[row].float().abs().max()) == 0.0 and float(dv_p[row].float().abs().max()) == 0.0
    # against the fp32 reference on the packed data (per sequence)
    if p == 0.0:
        from test_kernels_gpu import _attn_ref
        for b in range(B):
            nq, nk = min(qlens[b] + 1, Sq), klens[b]
            ref, _ = _attn_ref(q[b:b + 1, :nq], k[b:b + 1, :nk], v[b:b + 1, :nk], None, heads, scale)
            got = ctx_p[int(pq.off[b]):int(pq.off[b]) + nq].float().cpu()
            assert float((got - ref[0]).abs().max()) < 3e-2 * float(ref.abs().max()) + 1e-3


def test_attn_fused_packed_rejects_bad_arguments(ops):
    pq = _pack([5, 9], 9)
    vl = ops.AttnVarlen(pq, pq)
    q = torch.zeros(pq.M, 64, device="cuda", dtype=torch.bfloat16)
    with pytest.raises(ops.MMDTIError):
        ops.attn_fwd(q, q, q, torch.zeros(2, 9, device="cuda"), 2, 2, 9, 9, 0.1, vl=vl)       # key_add with packed sequences
    with pytest.raises(ops.MMDTIError):
        ops.attn_fwd(q[:-1], q[:-1], q[:-1], None, 2, 2, 9, 9, 0.1, vl=vl)                   # wrong row count


# ------------------------------------------------------------------------------------------------ InfoNCE mean / masked pool
@pytest.mark.parametrize("lens,S,D", [((5, 9, 1, 9), 9, 64), ((130, 17, 64), 130, 512), ((3,), 256, 512)])
def test_seq_mean_packed_is_the_unmasked_mean_over_padded_rows(ops, lens, S, D):
    """infonce.py:32-33 averages over ALL S positions; with identical pad rows that is (sum_real + n_pad * x_pad) / S."""
    B = len(lens)
    pk = _pack(lens, S)
    xp = rt(torch.randn(pk.M, D, generator=G(1)))
    padded = pk.unpack(xp)                                        # [B, S, D]: every padded slot holds the representative row
    m = ops.seq_mean_packed_fwd(dev(bf(xp)), pk, D, D)
    close(m, padded.double().mean(1).float(), 1e-5, 1e-6)
    m_dense = ops.seq_mean_fwd(dev(bf(padded)).view(B * S, D), B, S, D, D)
    close(m, m_dense, 1e-5, 1e-6)
    d = torch.randn(B, D, generator=G(2))
    u = rt(torch.randn(pk.M, D, generator=G(3)))
    w = (pk.pad_weights() / S).view(-1, 1)
    want = d[pk.seq_host] * w
    close(ops.seq_mean_packed_bwd(dev(d), pk, D, D).float().cpu(), rt(want), 1e-2, 1e-6)
    close(ops.seq_mean_packed_bwd(dev(d), pk, D, D, aux=dev(bf(u)), aux_mode=1).float().cpu(), want * u, 1e-2, 1e-5)
    # = the sum over the padded rows of the dense backward (every pad row carries the same gradient)
    dense = ops.seq_mean_bwd(dev(d), B, S, D, D).float().cpu().view(B, S, D)
    for b, n in enumerate(lens):
        if n < S:
            got = ops.seq_mean_packed_bwd(dev(d), pk, D, D).float().cpu()[int(pk.off_host[b]) + n]
            close(got, dense[b, n:].sum(0), 2e-2, 1e-6)


def test_masked_pool_packed(ops):
    B, Sa, St, D = 4, 9, 13, 64
    pa, pt = _pack([5, 9, 1, 9], Sa), _pack([13, 2, 7, 12], St)
    a, t = torch.randn(pa.M, D, generator=G(1)), torch.randn(pt.M, D, generator=G(2))
    ad, td = pa.unpack(a), pt.unpack(t)                       # padded forms (pad slots hold the representative rows)
    ma = torch.arange(Sa).view(1, -1) < pa.counts_host.view(-1, 1)
    mt = torch.arange(St).view(1, -1) < pt.counts_host.view(-1, 1)
    want = ops.masked_pool_fwd(dev(ad), dev(td), dev(ma).view(torch.uint8), dev(mt).view(torch.uint8))
    got = ops.masked_pool_packed_fwd(dev(a), dev(t), pa, pt)
    close(got, want, 1e-6, 1e-6)
    dp = torch.randn(B, D, generator=G(3))
    da, dt = ops.masked_pool_packed_bwd(dev(dp), pa, pt)
    da_d, dt_d = ops.masked_pool_bwd(dev(dp), dev(ma).view(torch.uint8), dev(mt).view(torch.uint8), Sa, St)
    assert torch.equal(da, da_d.view(B * Sa, D)[pa.gather]) and torch.equal(dt, dt_d.view(B * St, D)[pt.gather])
    for pk, dg in ((pa, da), (pt, dt)):                     # representative pad rows: zero gradient
        for b in range(B):
            if int(pk.counts_host[b]) < pk.S:
                assert float(dg[int(pk.off_host[b]) + int(pk.counts_host[b])].abs().max()) == 0.0


# ------------------------------------------------------------------------------------------------ whole step
def _small_refarch(task="classification", layers=2, **params):
    """The reference's widths (512 / 64 heads / 128 Gaussians; RoBERTa 512 / 8 heads; fusion 16 heads) at reduced depth: every
    kernel of the hot path -- the fused pair-bias kernels and the compact pair planes included -- at test cost."""
    ocfg = refarch_cfg(task, 600)
    ocfg.unimol.layers, ocfg.roberta.layers = layers, 2
    return ocfg, product_model(ocfg, **params).cuda()


def _host_fields(batch):
    from mmdti_hip.collate import device_payload
    full = device_payload(batch)
    return {k: full[k] for k in ("atom_counts", "token_counts", "token_pad_id", "packable")}


@pytest.mark.parametrize("task", ["classification", "regression"])
def test_packed_step_equals_padded_step_at_dropout_zero(task):
    """same weights, same mixed-length batch, dropout 0 -- the packed layout and the padded layout give the
    same losses (1e-6 relative; what differs is the order of two fp32 sums: the weighted InfoNCE mean and the masked pool) and the
    same gradients up to the run-to-run band of the atomics (2e-3 at the bottom of the network, ~1e-2 on the
    Gaussian tables whose sums cancel)."""
    from mmdti_hip.functional import CELossFn, MSELossFn
    ocfg, model = _small_refarch(task)
    P = O.init_params(ocfg, seed=5, std=0.05)
    load_fixture_weights(model, P)
    model.train()                                            # every dropout probability is 0
    batch, label = O.synth_batch(7, 40, 48, ocfg, seed=11, ragged=True)
    host = _host_fields(batch)
    assert host["packable"] and int(host["atom_counts"].min()) < batch["src_tokens"].shape[1]
    d = {k: v.cuda() for k, v in batch.items()}
    tgt = label.cuda().float() if task == "regression" else label.cuda().long()
    res = {}
    for layout in ("padded", "packed"):
        model.zero_grad(set_to_none=True)
        model.strict_reference = layout == "padded"
        logits, infonce, ct = model(**d, **host, return_infonce_loss=True, return_ct_loss=True, net_target=tgt)
        assert model.last_layout == layout
        tl = MSELossFn.apply(logits, tgt) if task == "regression" else CELossFn.apply(logits, tgt)
        loss = tl + 0.1 * infonce + 0.1 * ct
        loss.backward()
        torch.cuda.synchronize()
        res[layout] = dict(logits=logits.detach().clone(), infonce=float(infonce), ct=float(ct), loss=float(loss),
                           grads={n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None})
    a, b = res["padded"], res["packed"]
    for k in ("infonce", "ct", "loss"):
        # (the contrastive loss divides feature products by its temperature 0.07 before exp / log: the 1e-7 sum-order noise of the
        #  pooled features arrives ~ 14 x larger in it -- measured 1.1e-5 relative on the regression configuration)
        rel = 3e-5 if k == "ct" else 1e-6
        assert abs(a[k] - b[k]) <= rel * abs(a[k]) + 1e-7, (k, a[k], b[k])
    assert rel_l2(b["logits"], a["logits"]) < 1e-6
    assert set(a["grads"]) == set(b["grads"])
    worst = ("", 0.0)
    for n in a["grads"]:
        # (analytically zero gradients -- softmax shift invariance -- are rounding noise in both runs: test_g9_gpu.ZERO_GRADS)
        if float(a["grads"][n].abs().max()) < 1e-9 or any(z in n for z in ("pooler", "key.bias", "gbf_proj.linear2.bias")):
            continue
        r = rel_l2(b["grads"][n], a["grads"][n])
        worst = max(worst, (n, r), key=lambda t: t[1])
        lim = 4e-2 if n.startswith(("gbf.", "gbf_proj.")) else 6e-3
        assert r < lim, (n, r)
    print("packed vs padded: worst gradient", worst)


def test_packed_layout_is_chosen_on_the_host_and_falls_back(ops):
    ocfg, model = _small_refarch()
    model.eval()
    batch, label = O.synth_batch(4, 24, 30, ocfg, seed=3, ragged=True)
    d = {k: v.cuda() for k, v in batch.items()}
    host = _host_fields(batch)
    with torch.no_grad():
        ref = model(**d)                                                     # no host fields: padded
        assert model.last_layout == "padded"
        out = model(**d, **host)
        assert model.last_layout == "packed"
        close(out, ref, 1e-5, 1e-6)
        model(**d, **dict(host, packable=False))
        assert model.last_layout == "padded"
        model(**d, **dict(host, token_pad_id=7))                             # masked SMILES slots that do not hold the pad id
        assert model.last_layout == "padded"
        model.strict_reference = True
        model(**d, **host)
        assert model.last_layout == "padded"
        model.strict_reference = False
        full, _ = O.synth_batch(4, 24, 30, ocfg, seed=3, ragged=False)       # nothing padded: the padded layout is the packed one
        model(**{k: v.cuda() for k, v in full.items()}, **_host_fields(full))
        assert model.last_layout == "padded"


def test_packed_step_trains_with_dropout_on():
    """Dropout on (the reference's probabilities): the packed step runs, is finite, and its loss trajectory over a few optimizer
    steps tracks the padded layout's (equal in expectation, not bit for bit: the padded rows draw independent masks)."""
    from mmdti_hip.trainer import FineTuner
    ocfg = refarch_cfg("classification", 600)
    ocfg.unimol.layers, ocfg.roberta.layers = 2, 2
    batch, label = O.synth_batch(8, 40, 48, ocfg, seed=21, ragged=True)
    host = _host_fields(batch)
    d = dict({k: v.cuda() for k, v in batch.items()}, **host)
    traj = {}
    from mmdti_hip.runtime import dropout_state
    for layout in ("padded", "packed"):
        dropout_state.reseed(20240607)          # (the mask draws must not depend on how many seeds the tests before this one consumed)
        model = product_model(ocfg, dropout=True, strict_reference=layout == "padded").cuda().train()
        load_fixture_weights(model, O.init_params(ocfg, seed=5, std=0.05))
        tuner = FineTuner(model, "classification", learning_rate=2e-4, total_steps=100)
        outs = [tuner.step(d, label.cuda()) for _ in range(6)]
        assert model.last_layout == layout
        traj[layout] = [(float(o.loss), float(o.infonce_loss)) for o in outs]
        assert all(math.isfinite(x) for t in traj[layout] for x in t)
    # (8 molecules with dropout on: single steps scatter by ~20 % between two mask draws; the means over the steps agree)
    mean = {k: np.mean(np.array(v), axis=0) for k, v in traj.items()}
    assert np.all(np.abs(mean["padded"] - mean["packed"]) < 0.12 * np.abs(mean["padded"])), (mean, traj)
    for (la, ia), (lb, ib) in zip(traj["padded"], traj["packed"]):
        assert abs(la - lb) < 0.4 * abs(la) and abs(ia - ib) < 0.4 * abs(ia), traj


@pytest.mark.parametrize("B,N,lens", [(4, 130, (130, 37, 64, 5)), (3, 37, (37, 16, 17)), (2, 200, (33, 199))])
def test_gbf_bias_packed_rows_stop_at_the_representative_pad_row(ops, B, N, lens):
    """Packed token rows at the two ends of the pair chain (row_blocks): the fused pair-bias forward produces, for each molecule, only
    the 4-row query blocks up to its representative pad row (bit-identical there, nothing written behind them), and the complete
    backward visits only those -- with the gradient zero behind them all eight parameter gradients equal the ragged run's."""
    K, Fh, H, E = 128, 128, 64, 31 * 31
    ld, nt = ops.pair_ld(N), ops.pair_tiles(N)
    gen = G(41)
    dist = torch.rand(B, N, N, generator=gen) * 8
    et = torch.randint(1, E, (B, N, N), generator=gen)
    for b, n in enumerate(lens):
        dist[b, n:, :] = 0; dist[b, :, n:] = 0; et[b, n:, :] = 0; et[b, :, n:] = 0
    mul, bias = 1 + 0.1 * torch.randn(E, generator=gen), 0.1 * torch.randn(E, generator=gen)
    means, stds = torch.rand(K, generator=gen) * 3, torch.rand(K, generator=gen) * 3 - 1.5
    w1, b1 = dev(bf(torch.randn(Fh, K, generator=gen) * 0.2)), dev(torch.randn(Fh, generator=gen) * 0.1)
    w2, b2 = dev(bf(torch.randn(H, Fh, generator=gen) * 0.2)), dev(torch.randn(H, generator=gen) * 0.1)
    d = [dev(t) for t in (dist, et.to(torch.int16), mul, bias, means, stds)]
    kt = torch.tensor([(n + 15) // 16 for n in lens])
    ke = [ops.pair_key_tiles_effective(int(k), nt) for k in kt]
    rows = torch.tensor([min(n + 1, N) for n in lens])
    pre_f, pre_b, rb_f, rb_b = ops.gbf_tile_prefixes(kt, N, "cuda", rows)
    nb4 = (N + 3) // 4
    assert rb_f.tolist() == [min((int(r) + 3) // 4, nb4) for r in rows] and int(pre_f[-1]) == sum(min(4 * k, nb4) * min((int(r) + 3) // 4, nb4) for k, r in zip(ke, rows))
    ref_f, ref_b = ops.gbf_tile_prefixes(kt, N, "cuda")
    dense, _ = ops.gbf_bias_fwd(*d, w1, b1, w2, b2, ld, save=False, tiled=True, compact=True, tile_prefix=ref_f)
    canary = 123.0
    import mmdti_hip.ops as O_
    orig_empty = O_.pair_empty
    O_.pair_empty = lambda *a, **k: torch.full_like(orig_empty(*a, **k), canary)
    try:
        pk, _ = ops.gbf_bias_fwd(*d, w1, b1, w2, b2, ld, save=False, tiled=True, compact=True, tile_prefix=pre_f, row_blocks=rb_f)
    finally:
        O_.pair_empty = orig_empty
    un = lambda t: ops.pair_untile(t, N)
    for b in range(B):
        r4 = min(4 * int(rb_f[b]), N)                                         # rows produced (4-row granularity)
        kc = min(16 * ke[b], N)
        assert torch.equal(un(pk)[b, :, :r4, :kc], un(dense)[b, :, :r4, :kc])
        if r4 < N:
            assert bool((un(pk)[b, :, r4:, :kc] == canary).all())             # query rows past the representative one: not written
    g = ops.pair_tile(dev(torch.randn(B, H, N, N, generator=gen)), N, 0.0)
    gu = un(g).clone()
    for b in range(B):
        gu[b, :, int(rows[b]):, :] = 0.0                                       # what the packed attention backward leaves: zeros behind the
        gu[b, :, :, min(16 * ke[b], N):] = 0.0                                 # representative row and behind the kept key tiles
    g = ops.pair_tile(gu, N, 0.0)
    names = ("dw1", "db1", "dw2", "db2", "dmul", "dbias", "dmeans", "dstds")
    shapes = ((Fh, K), (Fh,), (H, Fh), (H,), (E,), (E,), (K,), (K,))
    got = {}
    for tag, pre, rb in (("ragged", ref_b, None), ("packed", pre_b, rb_b)):
        got[tag] = {n: torch.zeros(sh, device="cuda") for n, sh in zip(names, shapes)}
        gg = g.clone()
        if rb is not None:
            ggu = un(gg).clone()
            for b in range(B):
                ggu[b, :, min(4 * int(rb_b[b]), N):, :] = float("nan")          # never read
            gg = ops.pair_tile(ggu, N, 0.0)
        ops.gbf_bias_bwd_full(gg, *d, w1, b1, w2, ld, *[got[tag][n].view(-1) for n in names], tile_prefix=pre, row_blocks=rb)
    for n in names:
        a, b_ = got["packed"][n], got["ragged"][n]
        assert torch.isfinite(a).all(), n
        assert float((a - b_).norm() / (b_.norm() + 1e-12)) < 1e-5, n
