"""Pin the CPU oracle -- and the product's host-side collate -- to the reference's OWN execution of
``models/transformers.py``, ``models/mm_model.py`` and ``tasks/trainer.py`` (fixtures G9/G10, generated in the build
container by tests/golden/make_golden_g9.py atop tests/golden/ref_shims.py).

These fixtures pin the reference-owned wiring: the key-padding merge (transformers.py:122-135), bias chaining (:136-139),
x_norm before the final LN (:155-161), the 5-tuple (:183), MM_Model's tuple protocol (mm_model.py:585-618), the FDS
in-place aliasing (:579-581), the collate layout (:645-682) and the trainer's step / FDS pass / best-checkpoint logic
(tasks/trainer.py:142-328).  Uni-Core's own numerics stay "parity unpinned" (the layer inside G9 is the oracle's).
"""
import json

import numpy as np
import pytest
import torch

from oracle import mmdti_oracle as O
from oracle import trainer_oracle as TO


from g9util import T, samples_from, zero_dropout_cfg, tiny_cfg, refarch_cfg, tokenizer_from, product_model


def close(a, b, rtol=2e-5, atol=2e-6, msg=""):
    a = a.detach() if isinstance(a, torch.Tensor) else T(a)
    b = b.detach() if isinstance(b, torch.Tensor) else T(b)
    torch.testing.assert_close(a.double(), b.double(), rtol=rtol, atol=atol, msg=lambda m: f"{msg}: {m}")


def rel_l2(a, b):
    a, b = a.detach().double().flatten(), T(b).double().flatten() if not isinstance(b, torch.Tensor) else b.detach().double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


# ------------------------------------------------------------------------------------------------ G9a: encoder wiring
@pytest.mark.parametrize("tag", ["pad_nohead", "nopad_head", "pad_head"])
def test_g9_encoder_oracle(golden, tag):
    g = golden("g9_encoder_" + tag)
    P = {k[2:]: T(v).requires_grad_() for k, v in g.items() if k.startswith("w_")}
    H = int(g["heads"])
    cfg = O.UniMolCfg(layers=2, dim=64, ffn=128, heads=H)
    emb, bias0 = T(g["emb"]).requires_grad_(), T(g["bias0"]).requires_grad_()
    pm = T(g["padding_mask"]) if bool(g["has_padding"]) else None
    x, attn, delta, x_norm, delta_norm = O.unimol_encoder(emb, bias0, pm, P, cfg, pre="", training=False)
    close(x, g["x"], msg="x")
    close(attn, g["attn"], msg="attn (last layer's pre-softmax logits, -inf at padded keys)")
    close(delta, g["delta"], msg="delta_pair_repr")
    close(x_norm, g["x_norm"]); close(delta_norm, g["delta_norm"])
    # the reference's in-place key-padding merge, visible in the caller's tensor (:122-135)
    B, N = g["emb"].shape[:2]
    want = T(g["bias0"]).view(B, H, N, N).clone()
    if pm is not None:
        want.masked_fill_(pm.view(B, 1, 1, N), float("-inf"))
    assert torch.equal(want.view(B * H, N, N), T(g["attn_mask_after"]))
    names = sorted(P)
    gs = torch.autograd.grad((x * T(g["gx"])).sum(), [emb, bias0] + [P[n] for n in names], allow_unused=True)
    close(gs[0], g["d_emb"], atol=1e-6, msg="d_emb"); close(gs[1], g["d_bias"], atol=1e-6, msg="d_bias")
    for n, gp in zip(names, gs[2:]):
        assert (gp is not None) == bool(g["hasgrad_" + n]), n
        if gp is not None:
            close(gp, g["g_" + n], rtol=1e-4, atol=2e-6, msg=n)


# ------------------------------------------------------------------------------------------------ G9b: MM_Model wiring
def _fds_from(g, feature_dim, bucket_num):
    return O.FDSOracle(feature_dim, float(g["fds_min_value"]), float(g["fds_bin_width"]), bucket_num=bucket_num, bucket_start=0,
                       start_update=0, start_smooth=1, kernel="gaussian", ks=5, sigma=1, momentum=0.9)


@pytest.mark.parametrize("tag", ["cls", "reg_fds"])
def test_g9_model_tiny_oracle(golden, tag):
    g = golden("g9_model_tiny_" + tag)
    task = str(g["task"])
    P = {k[2:]: T(v).requires_grad_() for k, v in g.items() if k.startswith("w_") and T(v).is_floating_point()}
    vocab_rob = P["bert.embeddings.word_embeddings.weight"].shape[0]
    cfg = tiny_cfg(task, vocab_rob)
    samples = samples_from(g)
    batch = {k[2:]: T(v) for k, v in g.items() if k.startswith("b_") and k != "b_label"}
    label = T(g["b_label"])
    # a0: the collated layout (tokenizer output taken from the fixture)
    first = samples[:6]
    assert torch.equal(O.pad_1d_tokens([T(s[0]["src_tokens"]) for s in first], 0), batch["src_tokens"])
    assert torch.equal(O.pad_2d([T(s[0]["src_edge_type"]) for s in first], 0), batch["src_edge_type"])
    assert torch.equal(O.pad_2d([T(s[0]["src_distance"]) for s in first], 0.0), batch["src_distance"])
    fds, epoch = None, 0
    if task == "regression":
        fds = _fds_from(g, 64, 10)
        assert np.isclose(fds.min_value, O.FDSOracle.bins_from_raw(g["fds_raw"], 10, False)[0])
        labs = torch.cat([T(s[1]).view(1, 1) for s in samples]).float()
        for ep in (0, 1):
            fds.update_last_epoch_stats(ep)
            fds.update_running_stats(T(g[f"fds_feats_ep{ep}"]).clone(), labs, ep)
            for k, v in fds.state().items():
                close(v, g[f"fds_ep{ep}_{k}"], msg=f"FDS buffer {k} after epoch {ep}")
        fds.update_last_epoch_stats(2)
        for k, v in fds.state().items():
            close(v, g[f"fds_ep2_{k}"], msg=f"FDS buffer {k} at epoch 2")
        epoch = 2
    tgt = label.float() if task == "regression" else label.long()
    out = O.mm_forward(batch, P, cfg, net_target=tgt, fds=fds, epoch=epoch, training=True)
    loss, tl = O.step_loss(out, tgt, task)
    for k, ref in (("logits", "o_logits"), ("infonce", "o_infonce"), ("ct", "o_ct"), ("enc", "o_enc"), ("bert", "o_bert")):
        close(out[k], g[ref], rtol=1e-4, atol=1e-5, msg=k)
    close(tl, g["o_task_loss"], rtol=1e-4); close(loss, g["o_loss"], rtol=1e-4)
    # return protocol values: features returned with return_feature are the (FDS-smoothed, aliased) pooled features
    close(out["pooled"], g["r5_feats"], rtol=1e-4, atol=1e-5, msg="pooled/smoothed features (aliasing, mm_model.py:579-581)")
    close(out["logits"], g["r1_logits"], rtol=1e-4, atol=1e-5)
    assert list(g["arity"]) == [1, 2, 2, 2, 4] and bool(g["r6_is_tensor"])
    ev = O.mm_forward(batch, P, cfg, net_target=None, fds=fds, epoch=epoch, training=False)
    close(ev["logits"], g["r7_eval_logits"], rtol=1e-4, atol=1e-5, msg="eval logits (no FDS smoothing in eval)")
    names = [n for n in sorted(P) if f"hasgrad_{n}" in g and bool(g[f"hasgrad_{n}"])]
    gs = torch.autograd.grad(loss, [P[n] for n in names], allow_unused=True)
    worst = 0.0
    for n, gp in zip(names, gs):
        assert gp is not None, n
        if float(np.abs(g["g_" + n]).max()) > 1e-7:
            worst = max(worst, rel_l2(gp, g["g_" + n]))
    assert worst < 2e-3, worst
    frozen = [n for n in sorted(P) if f"hasgrad_{n}" in g and not bool(g[f"hasgrad_{n}"])]
    assert all(n.startswith("bert.pooler.") for n in frozen), frozen


@pytest.mark.parametrize("tag", ["cls", "reg"])
def test_g9_model_refarch_oracle(golden, tag):
    """The reference architecture (15L/512/64h, 6L RoBERTa, 16-head fusion) on a 4-molecule batch: weights come from the
    seed stored in the fixture."""
    g = golden("g9_model_refarch_" + tag)
    task = str(g["task"])
    cfg = refarch_cfg(task, int(g["vocab_rob"]))
    P = {k: v.requires_grad_() for k, v in O.init_params(cfg, seed=int(g["seed"]), std=float(g["std"])).items()}
    assert float(P["encoder.layers.7.fc1.weight"].detach()[5, 7]) == float(g["w_check"][0])      # same generator stream as the fixture
    batch = {k[2:]: T(v) for k, v in g.items() if k.startswith("b_") and k != "b_label"}
    label = T(g["b_label"])
    tgt = label.float() if task == "regression" else label.long()
    out = O.mm_forward(batch, P, cfg, net_target=tgt, training=True)
    loss, tl = O.step_loss(out, tgt, task)
    for k in ("logits", "infonce", "ct", "enc", "bert"):
        close(out[k], g["o_" + k], rtol=2e-4, atol=2e-5, msg=k)
    close(loss, g["o_loss"], rtol=1e-4)
    names = [str(n) for n in g["gn_names"]]
    gs = torch.autograd.grad(loss, [P[n] for n in names], allow_unused=True)
    for n, gp, ref in zip(names, gs, g["gn"]):
        assert gp is not None, n
        assert abs(float(gp.norm()) - float(ref)) <= 2e-3 * float(ref) + 1e-9, (n, float(gp.norm()), float(ref))
    for k in [k for k in g if k.startswith("g_")]:
        assert rel_l2(gs[names.index(k[2:])], g[k]) < 2e-3, k


# ------------------------------------------------------------------------------------------------ G9c: collate (a0)
def test_g9_collate_oracle_and_product(golden):
    """batch_collate_fn (mm_model.py:645-682): the oracle's restatement AND the product's host-side collate against the
    reference's own output -- bit-exact for every tensor, same key order, truncation and the label=None fallback."""
    g = golden("g9_collate")
    tok = tokenizer_from(str(g["tok_json"]), int(g["max_len"]))
    samples = samples_from(g)
    model = product_model(tiny_cfg("classification", len(tok)), tok)
    for name, fn in (("oracle", lambda s: TO.collate(s, 0, tok)), ("product", model.batch_collate_fn)):
        b, y = fn(samples)
        assert list(b.keys()) == [str(k) for k in g["key_order"]], name
        for k, v in b.items():
            ref = T(g["b_" + k])
            assert v.dtype == ref.dtype and torch.equal(v, ref), (name, k)
        assert torch.equal(y, T(g["label"])), name
        long_s = [(dict(s[0], smile=str(t)), s[1]) for s, t in zip(samples[:2], g["long_smiles"])]
        bl, _ = fn(long_s)
        assert torch.equal(bl["input_ids"], T(g["long_input_ids"])) and torch.equal(bl["attention_mask"], T(g["long_attention_mask"])), name
        _, none = fn([(samples[0][0], "a"), (samples[1][0], "b")])
        assert none is None and bool(g["label_none"]), name


# ------------------------------------------------------------------------------------------------ G10: trainer (a18)
@pytest.mark.parametrize("tag", ["reg_fds", "cls"])
def test_g10_trainer_oracle(golden, tag):
    g = golden("g10_trainer_" + tag)
    task = str(g["task"])
    hp = json.loads(str(g["hp_json"]))
    P = {k[3:]: T(v) for k, v in g.items() if k.startswith("w0_") and T(v).is_floating_point()}
    vocab_rob = P["bert.embeddings.word_embeddings.weight"].shape[0]
    cfg = tiny_cfg(task, vocab_rob)
    tok = tokenizer_from(str(g["tok_json"]), 38)
    train, valid = samples_from(g, "train_"), samples_from(g, "valid_")
    fds = None
    if task == "regression":
        mn, bw = O.FDSOracle.bins_from_raw(g["fds_raw"], 6, False)
        fds = O.FDSOracle(64, mn, bw, bucket_num=6, bucket_start=0, start_update=0, start_smooth=1, kernel="gaussian", ks=5, sigma=1,
                          momentum=0.9)
        metric, inc = (lambda y, p: float(np.mean((y - p) ** 2))), False
    else:
        from sklearn.metrics import roc_auc_score
        metric, inc = (lambda y, p: float(roc_auc_score(y.astype(int), p.astype(np.float32)))), True
    rec = {}
    torch.manual_seed(1234)
    y_pred, best, best_fds = TO.fit_predict(P, cfg, train, valid, tok, hp, fds=fds, increasing_metric=inc, metric=metric, record=rec)
    assert np.array_equal(np.array(rec["orders"]), g["batch_order"]), "loader passes / shuffles differ from the reference run"
    np.testing.assert_allclose(rec["steps"]["task"], g["step_task_loss"], rtol=2e-3, atol=2e-5)
    np.testing.assert_allclose(rec["steps"]["infonce"], g["step_infonce"], rtol=2e-3, atol=2e-5)
    np.testing.assert_allclose(rec["steps"]["ct"], g["step_ct"], rtol=5e-3, atol=5e-5)
    np.testing.assert_allclose(y_pred, g["y_pred"], rtol=2e-3, atol=2e-4)
    worst = max(rel_l2(best[n], g["ck_" + n]) for n in best if float(np.abs(g["ck_" + n]).max()) > 0)
    assert worst < 2e-3, worst
    if best_fds is not None:
        for k, v in best_fds.state().items():
            close(v, g["ck_FDS." + k], rtol=5e-3, atol=5e-5, msg="FDS." + k)
