"""GPU tests of the fp16 FORWARD-OPERAND mode (ops.set_forward_fp16 / MMDTI_FWD_FP16=1): every 16-bit tensor that feeds a forward GEMM
-- weights, LayerNorm / GELU / attention outputs, tower 1's q | k | v -- is fp16 (the reference's own AMP dtype,
tasks/trainer.py:181-182) instead of bf16; the backward keeps bf16 operands.  What it buys is the north star's "embeddings within
1e-3" of the reference's fp32 run, which bf16 operands cannot reach at 15 + 6 layers (profiles/r03_rounding_sites_fp16.json).

Kernel level: the fp16 instantiations against fp32 arithmetic on the SAME fp16-rounded operands (products of two fp16 values are
exact in fp32, so only the accumulation order differs).  Model level: the reference-architecture fixtures (the reference's own
run) with the embeddings held to 1e-3."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import mmdti_oracle as O
from g9util import T, refarch_cfg, product_model, load_fixture_weights, rel_l2, cosine, host_fields
from test_kernels_gpu import ops, dev, bf, rt, close, G, _pair_ref          # noqa: F401


@pytest.fixture
def fp16_mode(ops):
    was = ops.FWD_F16
    ops.set_forward_fp16(True); O.set_forward_fp16(True)
    yield
    ops.set_forward_fp16(was); O.set_forward_fp16(was)


def h16(t):
    return t.to(torch.float16)


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (130, 56, 72), (1000, 512, 512), (33, 1536, 512), (33280, 512, 512), (4096, 2048, 512)])
def test_gemm_with_fp16_operands(ops, M, N, K):
    g = G(M + N + K)
    x, w = h16(torch.randn(M, K, generator=g)), h16(torch.randn(N, K, generator=g) * 0.1)
    b = torch.randn(N, generator=g)
    ref = x.double() @ w.double().t() + b.double()
    for od in (torch.float32, torch.float16, torch.bfloat16):
        y = ops.linear_fwd(dev(x), dev(w), dev(b), out_dtype=od)
        assert y.dtype == od
        tol = {torch.float32: 2e-6, torch.float16: 1e-3, torch.bfloat16: 8e-3}[od]
        assert float((y.double().cpu() - ref).abs().max()) <= tol * float(ref.abs().max()) + 1e-6, od
    # fused epilogues on the fp16 path: GELU + saved gelu' (bf16), residual + dropout (fp32 out)
    if N % 8 == 0 and K % 64 == 0:
        u = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        a = ops.linear_fwd(dev(x), dev(w), dev(b), act=ops.ACT_GELU_FWD, aux_out=u)
        assert a.dtype == torch.float16
        close(a, torch.nn.functional.gelu(ref.float()), 2e-3, 2e-3)
        res = torch.randn(M, N, generator=g)
        y0 = ops.linear_fwd(dev(x), dev(w), dev(b), residual=dev(res), out_dtype=torch.float32)
        close(y0, (ref + res.double()).float(), 1e-5, 1e-5)
    with pytest.raises(ops.MMDTIError):
        ops.linear_fwd(dev(x), dev(w).bfloat16(), dev(b))                        # mixed operand types
    with pytest.raises(ops.MMDTIError):
        ops.linear_bwd_input(dev(x), dev(h16(torch.randn(K, 40, generator=g))))  # fp16 operands are built for forward shapes only


def test_casts_and_layernorm_in_fp16_mode(ops, fp16_mode):
    x = torch.randn(37, 512, generator=G(1)) * 3
    y = ops.cast_act16(dev(x))
    assert y.dtype == torch.float16 and torch.equal(y.cpu(), x.half())
    yb = ops.to_bf16(y)
    assert yb.dtype == torch.bfloat16 and torch.equal(yb.cpu(), x.half().float().bfloat16())
    view = ops.to_bf16(y[:, 128:256])                                              # a row-strided view
    assert torch.equal(view.cpu(), x.half()[:, 128:256].float().bfloat16())
    assert ops.to_bf16(yb) is yb
    gam, bet = dev(torch.rand(512, generator=G(2)) + 0.5), dev(torch.randn(512, generator=G(3)) * 0.1)
    _, h, m, r = ops.layernorm_fwd(dev(x), gam, bet, 1e-5)
    ref = torch.nn.functional.layer_norm(x, (512,), gam.cpu(), bet.cpu(), 1e-5)
    assert h.dtype == torch.float16
    close(h, ref, 1e-3, 1e-3)
    assert float((h.float().cpu() - ref).abs().mean()) < 2e-4                     # fp16: 8x finer than the bf16 output
    # the fused Linear + LayerNorm kernel on fp16 operands, fp16 LayerNorm output
    a, w = h16(torch.randn(300, 512, generator=G(4))), h16(torch.randn(512, 512, generator=G(5)) * 0.05)
    b, res = torch.randn(512, generator=G(6)), torch.randn(300, 512, generator=G(7))
    yf, _, hf, mean, rstd = ops.linear_ln_fwd(dev(a), dev(w), dev(b), gam, bet, 1e-5, residual=dev(res))
    y_ref = (a.double() @ w.double().t() + b.double() + res.double()).float()
    close(yf, y_ref, 1e-5, 1e-5)
    assert hf.dtype == torch.float16
    close(hf, torch.nn.functional.layer_norm(y_ref, (512,), gam.cpu(), bet.cpu(), 1e-5), 2e-3, 2e-3)


@pytest.mark.parametrize("B,N,H,lens,p", [(2, 130, 64, (130, 37), 0.0), (3, 70, 8, (70, 16, 33), 0.1), (1, 258, 8, (200,), 0.0)])
def test_pair_attn_forward_with_fp16_qkv(ops, B, N, H, lens, p):
    """fp16 q | k | v (compact planes, dense / ragged / packed rows): S and O against the fp32 reference on the fp16-rounded
    operands; the backward takes the bf16 rounding of the same tensor (ops.pair_attn_bwd converts)."""
    from mmdti_hip.packing import PackedRows
    D, ld, scale = H * 8, ops.pair_ld(N), 8 ** -0.5
    qkv = torch.randn(B, N, 3 * D, generator=G(1)).half()
    bias = torch.randn(B, H, N, N, generator=G(2))
    key_pad = torch.zeros(B, N, dtype=torch.bool)
    for b, n in enumerate(lens):
        key_pad[b, n:] = True
    bias_ld = torch.zeros(B, H, N, ld); bias_ld[..., :N] = bias
    b16 = ops.pair_tile(dev(bias_ld), N, float("-inf")).half()
    S, Oref, _, _ = _pair_ref(qkv.float(), b16.float().cpu().new_tensor(ops.pair_untile(b16, N).float().cpu()), key_pad, H, scale, torch.zeros(B, N, D), None)
    s16, o = ops.pair_attn_fwd(dev(qkv).view(B * N, 3 * D), b16, dev(key_pad), B, N, H, ld, scale)
    assert o.dtype == torch.float16 and s16.dtype == torch.float16
    fin = torch.isfinite(S.detach())
    close(ops.pair_untile(s16, N).float().cpu()[fin], S.detach()[fin], 2e-3, 2e-3)          # (the stored logits are fp16)
    assert float((o.view(B, N, D).float().cpu() - Oref.detach()).abs().max()) < 6e-3 * float(Oref.abs().max()) + 1e-3
    # against the bf16-operand kernel on the same values: the fp16 run is the closer one
    _, o_b = ops.pair_attn_fwd(dev(qkv.float().bfloat16()).view(B * N, 3 * D), b16, dev(key_pad), B, N, H, ld, scale)
    e16 = float((o.view(B, N, D).float().cpu() - Oref.detach()).abs().mean())
    eb = float((o_b.view(B, N, D).float().cpu() - Oref.detach()).abs().mean())
    assert e16 < 0.5 * eb, (e16, eb)
    # ragged + packed rows with dropout: equal to the fp16 dense run row for row
    kt = torch.tensor([(n + 15) // 16 for n in lens], dtype=torch.int32, device="cuda")
    kw = dict(drop_p=p, seed=9, site=4)
    s_d, o_d = ops.pair_attn_fwd(dev(qkv).view(B * N, 3 * D), b16, dev(key_pad), B, N, H, ld, scale, key_tiles=kt, **kw)
    pk = PackedRows(torch.tensor(lens), N, device="cuda")
    qp = dev(qkv).view(B * N, 3 * D)[pk.gather].contiguous()
    s_p, o_p = ops.pair_attn_fwd(qp, b16, dev(key_pad).view(-1)[pk.gather].contiguous(), B, N, H, ld, scale, key_tiles=kt, row_off=pk.off, **kw)
    assert o_p.dtype == torch.float16 and torch.equal(o_p, o_d[pk.gather])
    # backward through the conversion
    dO = dev(bf(torch.randn(pk.M, D, generator=G(5))))
    gz = torch.zeros_like(s_p, dtype=torch.float32)
    dq = ops.pair_attn_bwd(qp, s_p, dO, gz, B, N, H, ld, scale, True, key_tiles=kt, row_off=pk.off, **kw)
    gz2 = torch.zeros_like(gz)
    dq2 = ops.pair_attn_bwd(ops.to_bf16(qp), s_p, dO, gz2, B, N, H, ld, scale, True, key_tiles=kt, row_off=pk.off, **kw)
    assert dq.dtype == torch.bfloat16 and torch.equal(dq, dq2)


def _capture(model):
    store = {}
    real = model.encoder.encode

    def encode(*a, **k):
        out = real(*a, **k)
        store["enc"] = out[0].detach()
        return out

    model.encoder.encode = encode
    model.bert.register_forward_hook(lambda m, i, o: store.__setitem__("bert", o[0].detach()))
    return store


@pytest.mark.parametrize("layout", ["padded", "packed"])
@pytest.mark.parametrize("tag", ["cls", "reg"])
def test_g9_refarch_b32_embeddings_within_1e3_with_fp16_forward_operands(golden, tag, layout, fp16_mode):
    """The reference architecture (15 x 512 / 64 heads, 6-layer RoBERTa, fusion) against the reference's own fp32 run (fixture
    g9_model_refarch_b32_*), forward operands fp16: encoder_rep, out_bert, logits AND every loss within the north star's 1e-3;
    gradients (bf16 backward) inside the bands of the bf16 mode."""
    from mmdti_hip.functional import CELossFn, MSELossFn
    g = golden("g9_model_refarch_b32_" + tag)
    task = str(g["task"])
    ocfg = refarch_cfg(task, int(g["vocab_rob"]))
    P = O.init_params(ocfg, seed=int(g["seed"]), std=float(g["std"]))
    model = product_model(ocfg).cuda()
    load_fixture_weights(model, P)
    store = _capture(model)
    model.train()
    cpu = {k[2:]: T(v) for k, v in g.items() if k.startswith("b_") and k != "b_label"}
    batch = {k: v.cuda() for k, v in cpu.items()}
    if layout == "packed":
        batch.update(host_fields(cpu))
    label = T(g["b_label"]).cuda()
    tgt = label.float() if task == "regression" else label.long()
    logits, infonce, ct = model(**batch, return_infonce_loss=True, return_ct_loss=True, net_target=tgt, use_weight=False, epoch=0)
    assert model.last_layout == layout
    tl = MSELossFn.apply(logits, tgt) if task == "regression" else CELossFn.apply(logits, tgt)
    loss = 1.0 * tl + 0.1 * infonce + 0.1 * ct
    enc, bert = store["enc"], store["bert"]
    if layout == "packed":
        pk1, pk2 = model._pack_cache[3]
        enc, bert = pk1.unpack(enc), pk2.unpack(bert)
    r = dict(enc=rel_l2(enc[:8], g["o_enc"]), bert=rel_l2(bert[:8], g["o_bert"]), logits=rel_l2(logits, g["o_logits"]),
             infonce=abs(float(infonce) - float(g["o_infonce"])) / abs(float(g["o_infonce"])),
             task_loss=abs(float(tl) - float(g["o_task_loss"])) / abs(float(g["o_task_loss"])),
             loss=abs(float(loss) - float(g["o_loss"])) / abs(float(g["o_loss"])))
    loss.backward()
    grads = dict(model.named_parameters())
    full = {k[2:]: (rel_l2(grads[k[2:]].grad, g[k]), cosine(grads[k[2:]].grad, g[k])) for k in g if k.startswith("g_")}
    worst = max(full.items(), key=lambda t: t[1][0])
    r.update(worst_grad_rel_l2=worst[1][0], min_grad_cos=min(v[1] for v in full.values()))
    import json, os
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, f"parity_fp16_{tag}_{layout}.json"), "w") as f:
        json.dump(dict(r, worst_grad_param=worst[0]), f, indent=1)
    assert r["enc"] < 1e-3 and r["bert"] < 1e-3 and r["logits"] < 1e-3, r           # north star: embeddings within 1e-3 relative
    assert r["infonce"] < 1e-3 and r["loss"] < 1e-3 and r["task_loss"] < 1e-3, r    # ... and the losses
    assert worst[1][0] < 2.2e-2 and r["min_grad_cos"] > 0.99995, (worst, r)        # measured x 1.3: 7.7e-3 ... 1.64e-2 / 0.999966


def test_fp16_mode_trains(fp16_mode):
    """A few optimizer steps in the fp16 forward-operand mode: the fp16 weight shadow follows the optimizer (re-cast after every
    Adam step), the loss goes down, and the first step's loss equals the bf16 mode's to bf16-vs-fp16 rounding."""
    from mmdti_hip.trainer import FineTuner
    from mmdti_hip import ops as _ops
    ocfg = refarch_cfg("classification", 600)
    ocfg.unimol.layers, ocfg.roberta.layers = 2, 2
    batch, label = O.synth_batch(8, 24, 30, ocfg, seed=3, ragged=True)
    d = dict({k: v.cuda() for k, v in batch.items()}, **host_fields(batch))
    model = product_model(ocfg).cuda().train()
    load_fixture_weights(model, O.init_params(ocfg, seed=5, std=0.05))
    tuner = FineTuner(model, "classification", learning_rate=3e-4, total_steps=50, warmup_ratio=0.0)
    losses = [float(tuner.step(d, label.cuda()).loss) for _ in range(8)]
    assert all(math.isfinite(v) for v in losses) and losses[-1] < losses[0], losses
    w = model.encoder.layers[0].fc1.weight
    assert tuner.arena.shadow16 is not None
    from mmdti_hip.runtime import wfwd
    assert wfwd(w).dtype == torch.float16 and torch.equal(wfwd(w), w.detach().half())
    _ops.set_forward_fp16(False)
    model2 = product_model(ocfg).cuda().train()
    load_fixture_weights(model2, O.init_params(ocfg, seed=5, std=0.05))
    l0 = float(FineTuner(model2, "classification", learning_rate=3e-4, total_steps=50, warmup_ratio=0.0).step(d, label.cuda()).loss)
    _ops.set_forward_fp16(True)
    assert abs(l0 - losses[0]) < 3e-3 * abs(l0), (l0, losses[0])


# ------------------------------------------------------------------------------------------- round 4: no conversion pass
@pytest.mark.parametrize("rows", [33280, 4096 + 64 * 3, 8192 + 1, 1695, 513, 130])
def test_weight_gradients_convert_fp16_activations_inside_the_kernel(ops, rows):
    """dW += dy^T.x with x a saved fp16 activation (MMDTI_DT_B_F16 / x_f16): the tile is fetched as fp16 and rounded to bf16 between LDS
    and the matrix pipe.  The grouped kernels (slab split-K above 4096 rows, one workgroup per tile below) have no atomics, so the
    result must equal -- bit for bit -- the same launch on the separately converted tensor (ops.to_bf16: the pass this replaces)."""
    g = G(rows)
    shapes = [(512, 2048), (2048, 512), (1536, 512), (512, 512)]
    base = []
    for i, (no, ni) in enumerate(shapes):
        dy = bf(torch.randn(rows, no, generator=g)).cuda()
        wide = (torch.randn(rows, ni + 64, generator=g) * 3).half().cuda()
        x = wide[:, :ni] if i == 3 else wide[:, :ni].contiguous()        # (one row-strided view)
        base.append((dy, x, torch.randn(no, ni, generator=g).cuda(), torch.randn(no, generator=g).cuda() if i != 1 else None))

    def run(conv):
        items = [(dy, conv(x), dw.clone(), None if db is None else db.clone(), None) for dy, x, dw, db in base]
        ops.linear_bwd_weight_grouped(items)
        return [(it[2], it[3]) for it in items]

    a, b = run(lambda x: x), run(ops.to_bf16)
    for (dy, x, dw, db), (dwa, dba), (dwb, dbb) in zip(base, a, b):
        assert torch.equal(dwa, dwb), float((dwa - dwb).abs().max())
        if dba is not None:                     # (the bias gradient rides on dy alone; above 4096 rows its K-splits meet in fp32 atomics)
            close(dba, dbb, 1e-5, 1e-3) if rows > 4096 else None
            assert rows > 4096 or torch.equal(dba, dbb)
        close(dwa, dw + dy.float().t() @ x.float().bfloat16().float(), 2e-3, 2e-2)
    # the single-problem launch (fp32 atomics between K splits: equal up to their order) -- bare-load and ragged shapes
    for no, ni in ((512, 264), (1536, 512)):
        dy = bf(torch.randn(rows, no, generator=g)).cuda()
        x = (torch.randn(rows, ni, generator=g) * 3).half().cuda()
        dw, db = torch.ones(no, ni).cuda(), torch.ones(no).cuda()
        ops.linear_bwd_weight(dy, x, dw, db=db)
        close(dw, 1.0 + dy.float().t() @ x.float().bfloat16().float(), 2e-3, 2e-2)
        close(db, 1.0 + dy.float().sum(0), 2e-3, 2e-2)


@pytest.mark.parametrize("M,N,K", [(65536, 512, 2048), (65536, 2048, 512), (65536, 1536, 512)])
def test_gemm_256_tiles_with_fp16_operands(ops, M, N, K):
    """The 256 x 256 pipelined kernel's fp16 instantiation (tower 2's forward shapes) against the 128 x 128 kernels on the same operands."""
    from mmdti_hip import _abi
    lib = _abi.lib()
    g = G(M + N)
    x, w = h16(torch.randn(M, K, generator=g)).cuda(), h16(torch.randn(N, K, generator=g) * 0.1).cuda()
    b = torch.randn(N, generator=g).cuda()
    u = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    y_big = ops.linear_fwd(x, w, b, act=ops.ACT_GELU_FWD, aux_out=u)
    try:
        lib.mmdti_set_option(b"gemm_big", 0)
        u0 = torch.empty_like(u)
        y_128 = ops.linear_fwd(x, w, b, act=ops.ACT_GELU_FWD, aux_out=u0)
    finally:
        lib.mmdti_set_option(b"gemm_big", 1)
    assert y_big.dtype == torch.float16
    close(y_big, y_128, 2e-3, 2e-3)
    close(u, u0, 1e-2, 1e-2)
    ref = torch.nn.functional.gelu(x[:512].float() @ w.float().t() + b)
    close(y_big[:512], ref, 2e-3, 2e-3)


def test_fp16_stores_saturate_instead_of_overflowing(ops, fp16_mode):
    """Every fp16 epilogue clamps a value past the fp16 range to +-65504 (common.h f2h_sat2): an inf in a forward GEMM operand would
    turn the next product's whole row into NaN.  +-inf inputs clamp too; a NaN does not survive the packed min / max (the fp32
    stream written beside every 16-bit copy keeps it)."""
    big = 7.0e4
    # cast
    x = torch.tensor([[big, -big, 1.0, float("inf"), float("-inf"), float("nan"), 65504.0, 65520.0]]).repeat(4, 64).cuda()
    y = ops.cast_act16(x).float()
    assert y.dtype == torch.float32 and float(y[0, 0]) == 65504.0 and float(y[0, 1]) == -65504.0 and float(y[0, 2]) == 1.0
    assert float(y[0, 3]) == 65504.0 and float(y[0, 4]) == -65504.0 and math.isfinite(float(y[0, 5])) and float(y[0, 7]) == 65504.0
    # GEMM epilogue (fp16 out, + GELU) -- 128 x 128, small-tile and 256 x 256 kernels
    for M in (256, 2048, 65536):
        a = torch.full((M, 512), 16.0).half().cuda()
        w = torch.full((512, 512), 16.0).half().cuda()
        w[1::2] = -16.0                                                        # odd output columns: -131072
        out = ops.linear_fwd(a, w, None)
        assert out.dtype == torch.float16 and torch.isfinite(out).all()
        assert float(out[:, 0].min()) == 65504.0 and float(out[:, 1].max()) == -65504.0
        u = torch.empty(M, 512, device="cuda", dtype=torch.bfloat16)
        act = ops.linear_fwd(a, w, None, act=ops.ACT_GELU_FWD, aux_out=u)
        assert torch.isfinite(act).all() and float(act[:, 0].min()) == 65504.0 and float(act[:, 1].abs().max()) == 0.0
    # LayerNorm output and the fused Linear + LayerNorm
    gam, bet = torch.full((512,), 1.0e5).cuda(), torch.zeros(512).cuda()
    xs = torch.randn(300, 512, generator=G(1)).cuda()
    _, h, _, _ = ops.layernorm_fwd(xs, gam, bet, 1e-5)
    assert h.dtype == torch.float16 and torch.isfinite(h).all() and float(h.float().abs().max()) == 65504.0
    a, w = h16(torch.randn(300, 512, generator=G(2))).cuda(), h16(torch.randn(512, 512, generator=G(3)) * 0.05).cuda()
    _, _, hf, _, _ = ops.linear_ln_fwd(a, w, None, gam, bet, 1e-5, residual=xs)
    assert hf.dtype == torch.float16 and torch.isfinite(hf).all() and float(hf.float().abs().max()) == 65504.0
    # attention contexts: values at the edge of the range times the 1 / (1 - p) of dropout
    B, heads, L, hd = 2, 8, 64, 64
    q = bf(torch.randn(B * L, heads * hd, generator=G(4))).cuda()
    v = torch.full((B * L, heads * hd), 65280.0).bfloat16().cuda()
    ctx, _ = ops.attn_fwd(q, q, v, None, B, heads, L, L, 0.125, drop_p=0.5, seed=3, site=1)
    assert ctx.dtype == torch.float16 and torch.isfinite(ctx).all() and float(ctx.float().max()) == 65504.0
    Bm, N, H = 2, 40, 8
    qkv = torch.randn(Bm * N, 3 * H * 8, generator=G(5)).half().cuda()
    qkv[:, 2 * H * 8:] = 65504.0
    bias = ops.pair_tile(torch.zeros(Bm, H, N, ops.pair_ld(N)).cuda(), N).half()
    _, o = ops.pair_attn_fwd(qkv, bias, None, Bm, N, H, ops.pair_ld(N), 8 ** -0.5, drop_p=0.5, seed=3, site=2)
    assert o.dtype == torch.float16 and torch.isfinite(o).all() and float(o.float().max()) == 65504.0
    # the optimizer's fp16 weight shadow
    p = torch.tensor([6.55e4, -6.55e4, 1.0, 0.0] * 4).cuda()
    gr = torch.tensor([-1.0, 1.0, 0.0, 0.0] * 4).cuda()
    m, vv = torch.zeros_like(p), torch.zeros_like(p)
    pb, ph = torch.empty(16, device="cuda", dtype=torch.bfloat16), torch.empty(16, device="cuda", dtype=torch.float16)
    ops.adam_step(p, gr, m, vv, pb, 100.0, 0.9, 0.999, 1e-6, 0.0, 1, p_f16=ph)
    assert float(p[0]) > 65504.0 and float(ph[0]) == 65504.0 and float(ph[1]) == -65504.0 and float(ph[2]) == 1.0
    assert torch.equal(pb.float(), p.bfloat16().float())


def test_fp16_weight_shadow_follows_graph_replays(fp16_mode):
    """The Adam pass writes both 16-bit weight shadows through raw pointers, so a captured step's replays keep the fp16 one fresh: an
    eager evaluation after replays must read the weights of the last optimizer step (the shadow used to be re-cast by
    epoch bookkeeping the replay does not advance)."""
    from mmdti_hip.trainer import FineTuner
    from mmdti_hip.runtime import wfwd
    ocfg = refarch_cfg("classification", 600)
    ocfg.unimol.layers, ocfg.roberta.layers = 2, 2
    batch, label = O.synth_batch(8, 24, 30, ocfg, seed=3, ragged=False)
    d = {k: v.cuda() for k, v in batch.items()}
    model = product_model(ocfg).cuda().train()
    load_fixture_weights(model, O.init_params(ocfg, seed=5, std=0.05))
    tuner = FineTuner(model, "classification", learning_rate=1e-3, total_steps=50, warmup_ratio=0.0)
    tuner.step(d, label.cuda())
    for _ in range(3):
        tuner.graphed_step(d, label.cuda())
    torch.cuda.synchronize()
    for w in (model.encoder.layers[0].fc1.weight, model.encoder.layers[1].self_attn.in_proj.weight, list(model.bert.layers)[1].output.dense.weight):
        assert torch.equal(wfwd(w), w.detach().half())
