"""G9 / G10 fixtures: outputs of the reference's OWN ``models/transformers.py``, ``models/mm_model.py`` and
``tasks/trainer.py`` run on CPU in the build container (SURVEY.md 8c "G9").

    python tests/golden/make_golden_g9.py          # writes tests/golden/g9_*.npz, g10_*.npz

The three files import atop ``ref_shims.py`` (read its docstring: the ``unicore`` package is a stand-in built from the
oracle's restatement, so these fixtures pin the reference-owned wiring -- key-padding merge, bias chaining, x_norm
before the final LN, the 5-tuple, MM_Model's tuple protocol, FDS in-place aliasing, collate layout, the trainer's
loss mix / Adam / warm-up / FDS epoch pass / best-checkpoint reload -- not Uni-Core's numerics).  Fixtures are data only
(inputs, weights or the seed that regenerates them, outputs, gradients).  Nothing here runs on the GPU box.
"""
import importlib
import json
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shims as S          # noqa: E402
from ref_shims import O        # noqa: E402

OUT = HERE


def npz(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print("wrote", name, len(out), "arrays", f"{os.path.getsize(os.path.join(OUT, name + '.npz')) / 1024:.0f} KiB")


def ns(**kw):
    return types.SimpleNamespace(**kw)


# ------------------------------------------------------------------------------------------------ G9a: the encoder
def g9_encoder(T):
    cases = {
        "pad_nohead": dict(lens=[7, 5, 3], no_head_ln=True),          # MM_Model's configuration (no_final_head_layer_norm)
        "nopad_head": dict(lens=[6, 6], no_head_ln=False),            # padding_mask=None
        "pad_head": dict(lens=[9, 2, 9, 4], no_head_ln=False),
    }
    for tag, c in cases.items():
        torch.manual_seed(40)
        H, D = 8, 64
        enc = T.TransformerEncoderWithPair(encoder_layers=2, embed_dim=D, ffn_embed_dim=128, attention_heads=H, emb_dropout=0.1,
                                           dropout=0.1, attention_dropout=0.1, activation_dropout=0.0, max_seq_len=512,
                                           activation_fn="gelu", no_final_head_layer_norm=c["no_head_ln"]).eval()
        g = torch.Generator().manual_seed(41)
        with torch.no_grad():
            for n, p in enc.named_parameters():
                if "layer_norm" in n:
                    p.add_(0.1 * torch.randn(p.shape, generator=g))
                else:
                    p.copy_(0.08 * torch.randn(p.shape, generator=g))
        B, N = len(c["lens"]), max(c["lens"])
        emb = torch.randn(B, N, D, generator=g)
        pad = torch.zeros(B, N, dtype=torch.bool)
        for b, n in enumerate(c["lens"]):
            pad[b, n:] = True
            emb[b, n:] = 0.0
        padding_mask = pad if pad.any() else None
        bias0 = torch.randn(B * H, N, N, generator=g)
        emb_l = emb.clone().requires_grad_(True)
        bias_l = bias0.clone().requires_grad_(True)
        attn_mask = bias_l * 1.0                                   # non-leaf: the reference fills -inf into it IN PLACE
        x, attn, delta, x_norm, delta_norm = enc(emb_l, attn_mask=attn_mask, padding_mask=padding_mask)
        gx = torch.randn(x.shape, generator=g)
        params = list(enc.parameters())
        gs = torch.autograd.grad((x * gx).sum(), [emb_l, bias_l] + params, allow_unused=True, retain_graph=True)
        # the AUXILIARY outputs are differentiable in the reference (:141-181): gradients of x_norm, delta_pair_repr_norm and
        # delta_pair_repr (and of attn at its finite entries) with respect to the inputs and the parameters
        gd = torch.randn(delta.shape, generator=g)
        ga = torch.randn(attn.shape, generator=g)
        fin = torch.isfinite(attn)
        aux = 0.7 * x_norm + 1.3 * delta_norm + (delta * gd).sum() + (torch.where(fin, attn, torch.zeros_like(attn)) * ga).sum()
        gaux = torch.autograd.grad(aux, [emb_l, bias_l] + params, allow_unused=True)
        arrays = dict(emb=emb, bias0=bias0, padding_mask=pad, has_padding=np.array(padding_mask is not None), heads=H,
                      attn_mask_after=attn_mask, x=x, attn=attn, delta=delta, x_norm=x_norm, delta_norm=delta_norm, gx=gx,
                      d_emb=gs[0], d_bias=gs[1], g_delta=gd, g_attn=ga, d_emb_aux=gaux[0], d_bias_aux=gaux[1])
        for (n, p), gp, gq in zip(enc.named_parameters(), gs[2:], gaux[2:]):
            arrays["w_" + n] = p
            arrays["g_" + n] = torch.zeros_like(p) if gp is None else gp
            arrays["hasgrad_" + n] = np.array(gp is not None)
            arrays["gaux_" + n] = torch.zeros_like(p) if gq is None else gq
            arrays["hasgaux_" + n] = np.array(gq is not None)
        npz("g9_encoder_" + tag, **arrays)


# ------------------------------------------------------------------------------------------------ model construction
TINY = dict(mol=dict(encoder_layers=2, encoder_embed_dim=64, encoder_ffn_embed_dim=128, encoder_attention_heads=8),
            rob=dict(layers=2, dim=64, heads=4, ffn=128, max_pos=40),
            cross=dict(hidden_size=64, num_attention_heads=4, intermediate_size=128))
REFARCH = dict(mol=dict(encoder_layers=15, encoder_embed_dim=512, encoder_ffn_embed_dim=2048, encoder_attention_heads=64),
               rob=dict(layers=6, dim=512, heads=8, ffn=2048, max_pos=514),
               cross=dict(hidden_size=512, num_attention_heads=16, intermediate_size=2048))


def oracle_cfg(arch, task, vocab_rob, out_dim):
    m, r, c = arch["mol"], arch["rob"], arch["cross"]
    return O.ModelCfg(unimol=O.UniMolCfg(layers=m["encoder_layers"], dim=m["encoder_embed_dim"], ffn=m["encoder_ffn_embed_dim"],
                                         heads=m["encoder_attention_heads"], K=128, vocab=31),
                      roberta=O.RobertaCfg(layers=r["layers"], dim=r["dim"], heads=r["heads"], ffn=r["ffn"], vocab=vocab_rob,
                                           max_pos=r["max_pos"]),
                      cross=O.CrossCfg(dim=c["hidden_size"], heads=c["num_attention_heads"], ffn=c["intermediate_size"]),
                      task=task, output_dim=out_dim)


def build_reference_model(MM, arch, task, tmp, tok_json, vocab_rob, fds_raw=None, fds_num=10):
    """Construct the reference's MM_Model through its own ctor: synthetic mol.dict.txt + an empty Uni-Mol checkpoint +
    a locally saved random RoBERTa directory with the local tokenizer.  Dropout probabilities are configured to 0
    (value parity is defined at p = 0)."""
    from transformers import RobertaConfig, RobertaModel
    os.makedirs(tmp, exist_ok=True)
    with open(os.path.join(tmp, "mol.dict.txt"), "w") as f:
        f.write("\n".join(S.MOL_SYMBOLS) + "\n")
    unimol_dir = os.path.join(tmp, "mol_pre.pt")
    torch.save({"model": {}}, unimol_dir)
    r = arch["rob"]
    hf_cfg = RobertaConfig(vocab_size=vocab_rob, hidden_size=r["dim"], num_hidden_layers=r["layers"], num_attention_heads=r["heads"],
                           intermediate_size=r["ffn"], max_position_embeddings=r["max_pos"], type_vocab_size=1, pad_token_id=1,
                           bos_token_id=0, eos_token_id=2, layer_norm_eps=1e-12, hidden_dropout_prob=0.0,
                           attention_probs_dropout_prob=0.0)
    chem = os.path.join(tmp, "chemberta")
    RobertaModel(hf_cfg).save_pretrained(chem)
    S.fast_tokenizer(tok_json, r["max_pos"] - 2).save_pretrained(chem)

    m = arch["mol"]
    MM.molecule_architecture = lambda: ns(dropout=0.0, emb_dropout=0.0, attention_dropout=0.0, activation_dropout=0.0,
                                          pooler_dropout=0.0, max_seq_len=512, activation_fn="gelu", pooler_activation_fn="tanh",
                                          post_ln=False, backbone="transformer", kernel="gaussian", delta_pair_repr_norm_loss=-1.0, **m)
    c = arch["cross"]
    MM.crossmodal_config = lambda: ns(attention_probs_dropout_prob=0.0, gradient_checkpointing=False, hidden_act="gelu",
                                      hidden_dropout_prob=0.0, initializer_range=0.02, layer_norm_eps=1e-12,
                                      max_position_embeddings=512, num_hidden_layers=12, position_embedding_type="absolute", **c)
    MM.fds_config = lambda: ns(feature_dim=c["hidden_size"], bucket_num=20, bucket_start=0, start_update=0, start_smooth=1,
                               kernel="gaussian", ks=5, sigma=1, momentum=0.9, col_data="expt", raw_data="")
    params = dict(task=task, chemberta_dir=chem, unimol_dir=unimol_dir, ct_w=0.2)
    if fds_raw is not None:
        import pandas as pd
        csv = os.path.join(tmp, "train.csv")
        pd.DataFrame({"TARGET": fds_raw}).to_csv(csv, index=False)
        params.update(fds=True, fds_num=fds_num, fds_raw_path=csv, fds_col_data="TARGET", use_scaler=False)
    with S.cpu_device_moves():
        model = MM.MM_Model(output_dim=1 if task == "regression" else 2, **params)
    model.infonce.embed_dropout = 0.0
    return model


def load_oracle_params(model, P):
    missing, unexpected = model.load_state_dict({k: v.detach().clone() for k, v in P.items()}, strict=False)
    assert not unexpected, unexpected
    allowed = ("bert.pooler.", "FDS.", "bert.embeddings.position_ids", "bert.embeddings.token_type_ids")
    bad = [k for k in missing if not k.startswith(allowed)]
    assert not bad, bad


def synth_samples(n, max_atoms, tok_json, seed, task, with_weights=False):
    """DataHub-shaped samples: ({src_tokens, src_distance, src_coord, src_edge_type, smile[, weights]}, label)."""
    rng = np.random.default_rng(seed)
    alphabet = [c for c in json.loads(tok_json)["model"]["vocab"] if len(c) == 1]
    out = []
    for _ in range(n):
        na = int(rng.integers(3, max_atoms + 1))
        atoms = rng.choice(np.arange(4, 30), size=na)
        d = O.coords2unimol(atoms, rng.normal(0, 3.0, size=(na, 3)), 31)
        d["smile"] = "".join(rng.choice(alphabet, size=int(rng.integers(2, 2 * na + 2))))
        if with_weights:
            d["weights"] = float(rng.uniform(0.5, 1.5))
        if task == "regression":
            label = np.array([rng.normal(0, 1)], dtype=np.float32)
        else:
            label = np.array([int(rng.random() < 0.35)], dtype=np.int64)
        out.append((d, label))
    return out


def flat_samples(samples):
    arr = {"n_samples": len(samples), "smiles": np.array([s[0]["smile"] for s in samples])}
    for i, (d, y) in enumerate(samples):
        for k in ("src_tokens", "src_distance", "src_coord", "src_edge_type"):
            arr[f"s{i}_{k}"] = d[k]
        if "weights" in d:
            arr[f"s{i}_weights"] = d["weights"]
        arr[f"s{i}_label"] = y
    return arr


def run_step(model, batch, label, task, epoch, hooks):
    """One reference-protocol forward/backward (tasks/trainer.py:214-216 non-AMP branch)."""
    model.zero_grad()
    tgt = label.float() if task == "regression" else label.long()
    logits, infonce, ct = model(**batch, return_infonce_loss=True, return_ct_loss=True, net_target=tgt, use_weight=False, epoch=epoch)
    tl = torch.nn.functional.mse_loss(logits, tgt) if task == "regression" else \
        torch.nn.functional.cross_entropy(logits, tgt.flatten())
    loss = 1.0 * tl + 0.1 * infonce + 0.1 * ct
    loss.backward()
    return dict(logits=logits, infonce=infonce, ct=ct, task_loss=tl, loss=loss, enc=hooks["enc"], bert=hooks["bert"])


def attach_hooks(model):
    store = {}
    model.encoder.register_forward_hook(lambda m, i, o: store.__setitem__("enc", o[0].detach().clone()))
    model.bert.register_forward_hook(lambda m, i, o: store.__setitem__("bert", o[0].detach().clone()))
    return store


def g9_model(MM, tok_json, vocab_rob):
    # ---- tiny, classification (+ return protocol) and regression + FDS (aliasing, live smoothing at epoch 2)
    for task in ("classification", "regression"):
        with tempfile.TemporaryDirectory() as tmp:
            rng = np.random.default_rng(7)
            raw = rng.normal(0.0, 1.0, size=300)
            model = build_reference_model(MM, TINY, task, tmp, tok_json, vocab_rob, fds_raw=raw if task == "regression" else None)
        ocfg = oracle_cfg(TINY, task, vocab_rob, model.output_dim)
        P = O.init_params(ocfg, seed=90, std=0.08)
        load_oracle_params(model, P)
        hooks = attach_hooks(model)
        samples = synth_samples(12, 9, tok_json, seed=91, task=task)
        batches = [model.batch_collate_fn(samples[i:i + 6]) for i in (0, 6)]
        arrays = dict(task=np.array(task), **flat_samples(samples))
        for k, v in model.state_dict().items():
            if not k.startswith("FDS."):
                arrays["w_" + k] = v
        model.train()
        if task == "regression":
            arrays["fds_min_value"], arrays["fds_bin_width"] = model.FDS.min_value, model.FDS.bin_width
            arrays["fds_raw"] = raw
            # tasks/trainer.py:288-306 twice (end of epoch 0, end of epoch 1), then a training step at epoch 2
            for ep in (0, 1):
                feats, labs = [], []
                with torch.no_grad():
                    for b, y in batches:
                        _, f = model(**b, epoch=ep, return_feature=True, net_target=y.float())
                        feats.append(f)
                        labs.append(y.float())
                model.FDS.update_last_epoch_stats(ep)
                model.FDS.update_running_stats(torch.cat(feats), torch.cat(labs), ep)
                arrays[f"fds_feats_ep{ep}"] = torch.cat(feats)
                for k, v in model.FDS.state_dict().items():
                    arrays[f"fds_ep{ep}_{k}"] = v.clone()
            model.FDS.update_last_epoch_stats(2)
            for k, v in model.FDS.state_dict().items():
                arrays[f"fds_ep2_{k}"] = v.clone()
            epoch = 2
        else:
            epoch = 0
        b, y = batches[0]
        for k, v in b.items():
            arrays["b_" + k] = v
        arrays["b_label"] = y
        out = run_step(model, b, y, task, epoch, hooks)
        for k, v in out.items():
            arrays["o_" + k] = v
        for n, p in model.named_parameters():
            arrays["g_" + n] = torch.zeros_like(p) if p.grad is None else p.grad
            arrays["hasgrad_" + n] = np.array(p.grad is not None)
        # the return protocol (mm_model.py:585-618): arity and feature aliasing
        tgt = y.float() if task == "regression" else y.long()
        # (training + FDS + epoch >= start_smooth needs net_target: the reference iterates the labels, fds.py:164)
        nt = dict(net_target=tgt) if task == "regression" else {}
        with torch.no_grad():
            r1 = model(**b, epoch=epoch, **nt)
            r2 = model(**b, return_infonce_loss=True, epoch=epoch, **nt)
            r3 = model(**b, return_ct_loss=True, net_target=tgt, epoch=epoch)
            r4 = model(**b, return_feature=True, net_target=tgt, epoch=epoch)
            r5 = model(**b, return_feature=True, return_infonce_loss=True, return_ct_loss=True, net_target=tgt, epoch=epoch)
            model.eval()
            r6 = model(**b, return_ct_loss=True, epoch=epoch)                      # net_target None -> logits only
            r7 = model(**b, epoch=epoch)
            model.train()
        arrays.update(r1_logits=r1, r2_logits=r2[0], r2_infonce=r2[1], r3_logits=r3[0], r3_ct=r3[1], r4_logits=r4[0], r4_feats=r4[1],
                      r5_logits=r5[0], r5_feats=r5[1], r5_infonce=r5[2], r5_ct=r5[3], r6_is_tensor=np.array(torch.is_tensor(r6)),
                      r7_eval_logits=r7, arity=np.array([1, len(r2), len(r3), len(r4), len(r5)]))
        npz("g9_model_tiny_" + ("cls" if task == "classification" else "reg_fds"), **arrays)

    # ---- the reference architecture (15L/512/64h + 6L RoBERTa + 16-head fusion): weights regenerated from a seed
    for task in ("classification", "regression"):
        with tempfile.TemporaryDirectory() as tmp:
            model = build_reference_model(MM, REFARCH, task, tmp, tok_json, vocab_rob)
        ocfg = oracle_cfg(REFARCH, task, vocab_rob, model.output_dim)
        P = O.init_params(ocfg, seed=92, std=0.02)
        load_oracle_params(model, P)
        hooks = attach_hooks(model)
        model.train()
        samples = synth_samples(4, 20, tok_json, seed=93, task=task)
        b, y = model.batch_collate_fn(samples)
        out = run_step(model, b, y, task, 0, hooks)
        arrays = dict(task=np.array(task), seed=92, std=0.02, vocab_rob=vocab_rob, b_label=y,
                      w_check=np.array([float(P["encoder.layers.7.fc1.weight"][5, 7]), float(P["bert.encoder.layer.3.output.dense.weight"][1, 2])]))
        for k, v in b.items():
            arrays["b_" + k] = v
        for k, v in out.items():
            arrays["o_" + k] = v
        names, norms = [], []
        for n, p in model.named_parameters():
            if p.grad is not None:
                names.append(n)
                norms.append(float(p.grad.norm()))
        arrays["gn_names"], arrays["gn"] = np.array(names), np.array(norms)
        for n in ("classification_head.out_proj.weight", "encoder.layers.0.fc1.bias", "encoder.layers.14.self_attn.in_proj.bias",
                  "gbf.means.weight", "gbf.stds.weight", "encoder.emb_layer_norm.weight", "bert.embeddings.LayerNorm.weight",
                  "infonce.info_proj_query.2.weight", "gbf_proj.linear2.weight", "embed_tokens.weight"):
            arrays["g_" + n] = dict(model.named_parameters())[n].grad
        npz("g9_model_refarch_" + ("cls" if task == "classification" else "reg"), **arrays)


def g9_model_b32(MM, tok_json, vocab_rob):
    """The reference architecture at B = 32: a B x B InfoNCE softmax large enough that the mean over rows
    averages the per-row amplification of embedding rounding (at B = 4 the fixture above sits at 1.1-1.4e-3).  Tower outputs are
    stored for the first 8 molecules only (fixture size); losses, logits and gradient norms for the whole batch."""
    for task in ("classification", "regression"):
        with tempfile.TemporaryDirectory() as tmp:
            model = build_reference_model(MM, REFARCH, task, tmp, tok_json, vocab_rob)
        ocfg = oracle_cfg(REFARCH, task, vocab_rob, model.output_dim)
        P = O.init_params(ocfg, seed=92, std=0.02)
        load_oracle_params(model, P)
        hooks = attach_hooks(model)
        model.train()
        samples = synth_samples(32, 20, tok_json, seed=94, task=task)
        b, y = model.batch_collate_fn(samples)
        out = run_step(model, b, y, task, 0, hooks)
        arrays = dict(task=np.array(task), seed=92, std=0.02, vocab_rob=vocab_rob, b_label=y,
                      w_check=np.array([float(P["encoder.layers.7.fc1.weight"][5, 7]), float(P["bert.encoder.layer.3.output.dense.weight"][1, 2])]))
        for k, v in b.items():
            if k != "src_coord":                      # (never consumed: mm_model.py:540)
                arrays["b_" + k] = v
        for k, v in out.items():
            arrays["o_" + k] = v[:8] if k in ("enc", "bert") else v
        names, norms = [], []
        for n, p in model.named_parameters():
            if p.grad is not None:
                names.append(n)
                norms.append(float(p.grad.norm()))
        arrays["gn_names"], arrays["gn"] = np.array(names), np.array(norms)
        for n in ("classification_head.out_proj.weight", "encoder.layers.0.fc1.bias", "encoder.layers.14.self_attn.in_proj.bias",
                  "gbf.means.weight", "gbf.stds.weight", "encoder.emb_layer_norm.weight", "bert.embeddings.LayerNorm.weight",
                  "infonce.info_proj_query.2.weight", "gbf_proj.linear2.weight"):
            arrays["g_" + n] = dict(model.named_parameters())[n].grad
        npz("g9_model_refarch_b32_" + ("cls" if task == "classification" else "reg"), **arrays)


# ------------------------------------------------------------------------------------------------ G9c: collate
def g9_collate(MM, tok_json, vocab_rob):
    with tempfile.TemporaryDirectory() as tmp:
        model = build_reference_model(MM, TINY, "classification", tmp, tok_json, vocab_rob)
    samples = synth_samples(5, 11, tok_json, seed=95, task="regression", with_weights=True)
    b, y = model.batch_collate_fn(samples)
    arrays = dict(tok_json=np.array(tok_json), max_len=TINY["rob"]["max_pos"] - 2, key_order=np.array(list(b.keys())), label=y,
                  **flat_samples(samples))
    for k, v in b.items():
        arrays["b_" + k] = v
    # a SMILES longer than the tokenizer's model_max_length is truncated (truncation=True, mm_model.py:671)
    long_s = [(dict(s[0], smile=s[0]["smile"] * 9), s[1]) for s in samples[:2]]
    bl, _ = model.batch_collate_fn(long_s)
    arrays["long_smiles"] = np.array([s[0]["smile"] for s in long_s])
    arrays["long_input_ids"], arrays["long_attention_mask"] = bl["input_ids"], bl["attention_mask"]
    # labels that cannot be stacked -> label None (the bare except at :676-679)
    _, lab_none = model.batch_collate_fn([(samples[0][0], "a"), (samples[1][0], "b")])
    arrays["label_none"] = np.array(lab_none is None)
    npz("g9_collate", **arrays)


# ------------------------------------------------------------------------------------------------ G10: the trainer
def g10_trainer(MM, TR, tok_json, vocab_rob):
    for task in ("regression", "classification"):
        with tempfile.TemporaryDirectory() as tmp:
            rng = np.random.default_rng(17)
            raw = rng.normal(0.0, 1.0, size=200)
            fds = task == "regression"
            model = build_reference_model(MM, TINY, task, tmp, tok_json, vocab_rob, fds_raw=raw if fds else None, fds_num=6)
            ocfg = oracle_cfg(TINY, task, vocab_rob, model.output_dim)
            P = O.init_params(ocfg, seed=96, std=0.08)
            load_oracle_params(model, P)
            train = synth_samples(22, 9, tok_json, seed=97, task=task)      # 22 -> 5 batches of 4, drop_last drops 2
            valid = synth_samples(7, 9, tok_json, seed=98, task=task)       # last validation batch is short (3)
            tds = [(d, y) for d, y in train]
            vds = [(d, y) for d, y in valid]
            arrays = dict(task=np.array(task), tok_json=np.array(tok_json), fds_raw=raw)
            arrays.update({"train_" + k: v for k, v in flat_samples(train).items()})
            arrays.update({"valid_" + k: v for k, v in flat_samples(valid).items()})
            for k, v in model.state_dict().items():
                if not k.startswith("FDS."):
                    arrays["w0_" + k] = v.clone()
            hp = dict(task=task, metrics="mse" if task == "regression" else "auc", seed=42, learning_rate=5e-4, batch_size=4, epochs=4,
                      warmup_ratio=0.1, patience=10, max_norm=5.0, use_cuda=False, use_amp=False, alpha=1, beta=0.1, fds=fds)
            trainer = TR.Trainer(save_path=tmp, **hp)
            arrays["hp_json"] = np.array(json.dumps(hp))
            if task == "regression":
                base_loss = torch.nn.MSELoss()
                act = lambda x: x
            else:
                def base_loss(o, t):                     # models/loss.py:278-289 myCrossEntropyLoss (flattens [B,1] targets)
                    return torch.nn.functional.cross_entropy(o, t.flatten().long())
                act = lambda x: torch.nn.functional.softmax(x, dim=-1)[:, 1:]
            rec = {"task": [], "order": []}

            def loss_func(o, t):
                v = base_loss(o, t)
                if model.training and torch.is_grad_enabled():
                    rec["task"].append(float(v))
                return v

            real_collate = model.batch_collate_fn
            ids = {id(d): i for i, (d, _) in enumerate(train)}
            ids.update({id(d): 100 + i for i, (d, _) in enumerate(valid)})

            def collate(samples):
                # phase 0: training step, 1: the FDS statistics pass (train mode, no_grad), 2: validation / prediction
                phase = 2 if not model.training else (0 if torch.is_grad_enabled() else 1)
                rec["order"].append([phase] + [ids[id(s[0])] for s in samples] + [-1] * (4 - len(samples)))
                return real_collate(samples)

            model.batch_collate_fn = collate
            step_out = []
            real_forward = model.forward

            def forward(*a, **k):
                r = real_forward(*a, **k)
                if k.get("return_infonce_loss") and k.get("return_ct_loss"):
                    step_out.append((float(r[1]), float(r[2])))
                return r

            model.forward = forward
            torch.manual_seed(1234)                      # the DataLoader's shuffle draws from the global generator
            with S.cpu_device_moves():
                y_pred = trainer.fit_predict(model, tds, vds, loss_func, act, tmp, 0, None, return_infonce_loss=True,
                                             return_ct_loss=True, use_weight=False)
            ck = torch.load(os.path.join(tmp, "model_0.pth"), map_location="cpu")["model_state_dict"]
        # every training step calls loss_func twice in the non-AMP branch (:216-220): keep one per step
        arrays["step_task_loss"] = np.array(rec["task"][0::2])
        arrays["step_infonce"] = np.array([s[0] for s in step_out])
        arrays["step_ct"] = np.array([s[1] for s in step_out])
        arrays["batch_order"] = np.array(rec["order"])
        arrays["y_pred"] = y_pred
        for k, v in ck.items():
            arrays["ck_" + k] = v
        for k, v in model.state_dict().items():
            arrays["w1_" + k] = v
        npz("g10_trainer_" + ("reg_fds" if task == "regression" else "cls"), **arrays)


if __name__ == "__main__":
    S.install()
    T = importlib.import_module("models.transformers")
    MM = importlib.import_module("models.mm_model")
    TR = importlib.import_module("tasks.trainer")
    tok_json, vocab_rob = S.make_tokenizer_json()
    which = sys.argv[1:] or ["encoder", "model", "model_b32", "collate", "trainer"]
    if "encoder" in which:
        g9_encoder(T)
    if "model" in which:
        g9_model(MM, tok_json, vocab_rob)
    if "model_b32" in which:
        g9_model_b32(MM, tok_json, vocab_rob)
    if "collate" in which:
        g9_collate(MM, tok_json, vocab_rob)
    if "trainer" in which:
        g10_trainer(MM, TR, tok_json, vocab_rob)
