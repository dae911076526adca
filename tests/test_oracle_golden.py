"""Pin the CPU oracle against golden vectors produced by the reference itself
(tests/golden/make_golden.py)."""
import numpy as np
import pytest
import torch

from oracle import mmdti_oracle as O


def T(a):
    return torch.from_numpy(np.asarray(a))


def close(a, b, rtol=1e-5, atol=1e-6):
    a = a.detach() if isinstance(a, torch.Tensor) else T(a)
    b = b.detach() if isinstance(b, torch.Tensor) else T(b)
    torch.testing.assert_close(a.double(), b.double(), rtol=rtol, atol=atol)


# ---------------------------------------------------------------- G1 / G2
@pytest.mark.parametrize("B", [2, 16])
def test_g1_info_nce(golden, B):
    g = golden(f"g1_info_nce_B{B}")
    q, k = T(g["q"]).requires_grad_(), T(g["k"]).requires_grad_()
    loss = O.info_nce(q, k, temperature=0.1)
    dq, dk = torch.autograd.grad(loss, (q, k))
    close(loss, g["loss"]); close(dq, g["dq"]); close(dk, g["dk"])


def test_g1_info_nce_errors():
    q = torch.randn(4, 50)
    with pytest.raises(ValueError):
        O.info_nce(q[0], q)
    with pytest.raises(ValueError):
        O.info_nce(q, q[:, :10])
    with pytest.raises(ValueError):
        O.info_nce(q, q[:3])
    with pytest.raises(ValueError):
        O.info_nce(q, q, torch.randn(6, 50), negative_mode="paired")
    # explicit negatives: the reference's symmetric CE raises for non-square logits (infonce.py:98)
    with pytest.raises((ValueError, RuntimeError)):
        O.info_nce(q, q, torch.randn(6, 50), negative_mode="unpaired")


@pytest.mark.parametrize("mode", ["eval", "train_p0"])
def test_g2_infonce_module(golden, mode):
    g = golden(f"g2_infonce_module_{mode}")
    P = {"infonce." + k[2:]: T(v).requires_grad_() for k, v in g.items() if k.startswith("w_")}
    xq, xk = T(g["xq"]).requires_grad_(), T(g["xk"]).requires_grad_()
    loss = O.infonce_forward(xq, xk, P, p=0.0, training=(mode != "eval"))
    names = sorted(P)
    gs = torch.autograd.grad(loss, [xq, xk] + [P[n] for n in names])
    close(loss, g["loss"]); close(gs[0], g["dxq"], atol=1e-7); close(gs[1], g["dxk"], atol=1e-7)
    for n, gp in zip(names, gs[2:]):
        close(gp, g["g_" + n[len("infonce."):]], atol=1e-7)


# ---------------------------------------------------------------- G3
def _cases(g):
    names = sorted({k.split("__")[0] for k in g})
    return {n: {k.split("__")[1]: v for k, v in g.items() if k.startswith(n + "__")} for n in names}


def test_g3_contrastive(golden):
    cases = _cases(golden("g3_contrastive"))
    assert len(cases) >= 20
    for name, c in cases.items():
        f = T(c["f"]).requires_grad_()
        y = T(c["y"])
        wts = T(c["wts"]) if "wts" in c and bool(c.get("use_w", False)) else None
        if name.startswith("regress"):
            loss = O.ct_regress(f, y, T(c["yhat"]), weights=wts, w=float(c["w"]))
        elif name.startswith("single"):
            loss = O.ct_single(f, y, None, weights=wts)
        else:
            loss = O.ct_multi(f, y, None, weights=wts)
        (df,) = torch.autograd.grad(loss, f, allow_unused=True)
        df = torch.zeros_like(f) if df is None else df
        close(loss, c["loss"], rtol=2e-5), name
        close(df, c["df"], rtol=2e-4, atol=1e-6), name


# ---------------------------------------------------------------- G4
def test_g4_calibrate(golden):
    g = golden("g4_calibrate")
    x, m1, v1, m2, v2 = (T(g[k]) for k in ("x", "m1", "v1", "m2", "v2"))
    close(O.calibrate_mean_var(x.clone(), m1, v1, m2, v2), g["out_full"])
    close(O.calibrate_mean_var(x.clone(), m1, T(g["v1z"]), m2, v2), g["out_part"])
    close(O.calibrate_mean_var(x.clone(), m1, torch.zeros(8), m2, v2), g["out_tiny"])


@pytest.mark.parametrize("tag", ["gauss51", "gauss52_bs2", "triang", "laplace"])
def test_g4_fds_trajectory(golden, tag):
    g = golden(f"g4_fds_{tag}")
    kw = dict(bucket_num=int(g["cfg_bucket_num"]), bucket_start=int(g["cfg_bucket_start"]), kernel=str(g["cfg_kernel"]),
              ks=int(g["cfg_ks"]), sigma=float(g["cfg_sigma"]))
    mn, bw = O.FDSOracle.bins_from_raw(g["raw"], kw["bucket_num"], bool(g["cfg_using_scale"]))
    assert mn == pytest.approx(float(g["min_value"]), rel=1e-12)
    assert bw == pytest.approx(float(g["bin_width"]), rel=1e-12)
    f = O.FDSOracle(16, mn, bw, **kw)
    close(f.kernel_window, g["window"])
    lab, feats0, xb = T(g["labels"]), T(g["feats0"]), T(g["xb"])
    # integer part: bucket ids bit-exact
    assert torch.equal(O.fds_label_bins(lab, mn, bw), T(g["label_bin"]).long())

    def check(stage):
        for k, v in f.state().items():
            close(v, g[f"{stage}_{k}"], rtol=1e-5, atol=1e-6)

    f.update_last_epoch_stats(0)
    f.update_running_stats(feats0.clone(), lab, 0)
    check("s0")
    f.update_last_epoch_stats(1)
    check("s1")
    close(f.smooth(xb.clone(), lab[:40], 1), g["smooth1"], rtol=1e-5, atol=1e-5)
    close(f.smooth(xb.clone(), lab[:40], 0), g["smooth0"])
    f.update_running_stats((feats0 * 0.7 + 0.1).clone(), lab, 1)
    check("s2")
    f.update_last_epoch_stats(2)
    close(f.smooth(xb.clone(), lab[:40], 2), g["smooth2"], rtol=1e-5, atol=1e-5)
    check("s3")


# ---------------------------------------------------------------- G5
@pytest.mark.parametrize("tag", ["d64h4", "d128h4"])
def test_g5_cross_encoder(golden, tag):
    g = golden(f"g5_cross_{tag}")
    P = {k[2:]: T(v).requires_grad_() for k, v in g.items() if k.startswith("w_")}
    s1, s2 = T(g["s1"]).requires_grad_(), T(g["s2"]).requires_grad_()
    D = s1.shape[-1]
    cfg = O.CrossCfg(dim=D, heads=int(g["heads"]), ffn=P["layer.0.intermediate.dense.weight"].shape[0])
    add = (1.0 - T(g["mask2"])) * -10000.0
    out = O.cross_layer(s1, s2, add, P, "layer.0.", cfg)
    close(out, g["out"], rtol=1e-4, atol=1e-5)
    names = sorted(P)
    gs = torch.autograd.grad((out * T(g["gout"])).sum(), [s1, s2] + [P[n] for n in names])
    close(gs[0], g["ds1"], rtol=1e-4, atol=1e-5); close(gs[1], g["ds2"], rtol=1e-4, atol=1e-5)
    for n, gp in zip(names, gs[2:]):
        if "g_" + n in g:
            close(gp, g["g_" + n], rtol=1e-4, atol=1e-5)


# ---------------------------------------------------------------- G6
@pytest.mark.parametrize("impl", ["eager", "sdpa"])
def test_g6_roberta(golden, impl):
    g = golden(f"g6_roberta_{impl}")
    P = {"bert." + k[2:]: T(v).requires_grad_() for k, v in g.items() if k.startswith("w_")}
    ids, am = T(g["input_ids"]), T(g["attention_mask"])
    # integer part: position ids bit-exact, incl. a pad in the middle of a sequence
    assert torch.equal(O.roberta_position_ids(ids, 1), T(g["position_ids"]))
    cfg = O.RobertaCfg(layers=2, dim=32, heads=int(g["heads"]), ffn=64, vocab=40, max_pos=24, pad_idx=1)
    out = O.roberta_encoder(ids, am, P, cfg)
    close(out, g["out"], rtol=1e-4, atol=1e-5)
    names = [n for n in sorted(P) if "pooler" not in n]
    gs = torch.autograd.grad((out * T(g["gout"])).sum(), [P[n] for n in names], allow_unused=True)
    for n, gp in zip(names, gs):
        ref = g["g_" + n[len("bert."):]]
        gp = torch.zeros_like(P[n]) if gp is None else gp
        close(gp, ref, rtol=2e-4, atol=2e-5)
    # the pooler gets no gradient in the reference's usage either
    assert not bool(g["hasgrad_pooler.dense.weight"])


# ---------------------------------------------------------------- G7 / G8
def test_g7_pad(golden):
    g = golden("g7_pad")
    toks = [T(g[f"tok{i}"]) for i in range(3)]
    d2 = [T(g[f"d{i}"]) for i in range(3)]
    co = [T(g[f"c{i}"]) for i in range(3)]
    assert torch.equal(O.pad_1d_tokens(toks, 0), T(g["pad1d"]))
    assert torch.equal(O.pad_2d(d2, 0.0), T(g["pad2d"]))
    assert torch.equal(O.pad_coords(co, 0.0), T(g["padc"]))


def test_g8_gaussian(golden):
    g = golden("g8_gaussian")
    assert O.GBF_A == pytest.approx(2.5066272160, abs=1e-9)
    v = O.gaussian(T(g["x"]).float()[:, None], T(g["mean"]).float(), T(g["std"]).float())
    close(v, g["val"], rtol=1e-5, atol=1e-7)
    assert float(O.gaussian(torch.tensor(1.5), torch.tensor(1.0), torch.tensor(0.5))) == pytest.approx(0.48394164, rel=1e-6)


def test_edge_type_and_layout():
    d = O.coords2unimol(np.array([4, 8, 8, 5]), np.random.default_rng(0).normal(size=(4, 3)), vocab=31)
    t = d["src_tokens"]
    assert list(t) == [1, 4, 8, 8, 5, 2]
    assert d["src_edge_type"][2, 4] == 8 * 31 + 5
    assert d["src_distance"][0, 5] == 0.0          # BOS and EOS both sit at the origin
    assert np.allclose(d["src_distance"], d["src_distance"].T)


def test_full_step_runs_and_is_consistent():
    cfg = O.ModelCfg(unimol=O.UniMolCfg(layers=2, dim=64, ffn=128, heads=8, K=16, vocab=31),
                     roberta=O.RobertaCfg(layers=1, dim=64, heads=4, ffn=128, vocab=40, max_pos=40),
                     cross=O.CrossCfg(dim=64, heads=4, ffn=128), task="classification", output_dim=2)
    P = {k: v.requires_grad_() for k, v in O.init_params(cfg, seed=3).items()}
    batch, label = O.synth_batch(6, 9, 12, cfg, seed=5, ragged=True)
    out = O.mm_forward(batch, P, cfg, net_target=label)
    loss, tl = O.step_loss(out, label, cfg.task)
    loss.backward()
    assert torch.isfinite(loss)
    assert P["bert.embeddings.word_embeddings.weight"].grad is not None
    # padded key columns never leak: changing a padded atom's distance row must not change anything
    b2 = {k: v.clone() for k, v in batch.items()}
    pad = b2["src_tokens"].eq(0)
    assert pad.any()
    b2["src_distance"][pad.unsqueeze(1).expand_as(b2["src_distance"])] = 7.0
    out2 = O.mm_forward(b2, P, cfg, net_target=label)
    torch.testing.assert_close(out2["logits"], out["logits"])
