"""The reference's CONSTRUCTOR path on the product model (SURVEY 8b "construct").

``finetune.py`` -> ``NNModel._init_model`` (models/nnmodel.py:98-116) builds ``MM_Model(output_dim, **params)`` from two paths on disk
(models/mm_model.py:409-435, 472-476, 499-514): ``unimol_dir`` -- a ``torch.save({'model': state_dict})`` file whose SIBLING
``mol.dict.txt`` is the atom dictionary -- and ``chemberta_dir`` -- a HuggingFace directory (``config.json`` + weights + tokenizer).
Every other test builds through ``MM_Model.from_configs``; here the files are written to ``tmp_path`` and the product runs
``Dictionary.load``, the strict tower-1 key check of ``load_pretrained_weights``, ``RobertaTower.from_pretrained`` on a directory
written by HF's own ``save_pretrained``, ``AutoTokenizer.from_pretrained`` and the tokenizer hand-off in ``batch_collate_fn`` --
then (GPU half) one ``Trainer.fit_predict`` epoch and ``predict(load_model=True)`` exactly as ``NNModel.run`` / ``evaluate`` do
(models/nnmodel.py:213-232).  The tower-1 architecture is the reference's fixed one (15 x 512 / 64 heads); tower 2 is a one-layer
RoBERTa of hidden size 512 (what the fusion block requires); weights are random: the point is the path, not the numbers."""
import json
import os

import numpy as np
import pytest
import torch

from g9util import samples_from, T

MOL_SYMBOLS = ("[PAD] [CLS] [SEP] [UNK] C N O S H Cl F Br I Si P B Na K Al Ca Sn As Hg Fe Zn Cr Se Gd Au Li").split()


def _write_inputs(tmp_path, golden, drop_key=None):
    """-> (unimol_dir, chemberta_dir, tower-1 state dict saved, HF RobertaModel saved)."""
    from tokenizers import Tokenizer
    from transformers import PreTrainedTokenizerFast, RobertaConfig, RobertaModel
    from mmdti_hip.models import mm_model as mm
    g = golden("g9_collate")
    udir = tmp_path / "unimol"
    udir.mkdir(exist_ok=True)
    with open(udir / "mol.dict.txt", "w", encoding="utf-8") as f:
        for i, s in enumerate(MOL_SYMBOLS):
            f.write(f"{s} {1000 - i}\n")                       # "symbol count" lines, as unicore's Dictionary files are
    # a tower-1 checkpoint in the reference's format: {'model': state_dict}, keys as models/mm_model.py names them
    torch.manual_seed(7)
    donor = mm.MM_Model.from_configs(2, "classification", roberta_cfg=_tiny_roberta_ns(600))
    tower1 = {k: (v.clone().normal_(0, 0.03) if v.is_floating_point() else v.clone()) for k, v in donor.state_dict().items()
              if k.startswith(("embed_tokens.", "encoder.", "gbf.", "gbf_proj."))}
    saved = {k: v for k, v in tower1.items() if k != drop_key}
    saved["lm_head.extra"] = torch.zeros(3)                  # (pre-training heads the fine-tune model does not have: ignored, as strict=False does)
    torch.save({"model": saved}, udir / "mol_pre_no_h_220816.pt")
    cdir = tmp_path / "chemberta"
    tok = PreTrainedTokenizerFast(tokenizer_object=Tokenizer.from_str(str(g["tok_json"])), bos_token="<s>", eos_token="</s>", pad_token="<pad>",
                                  unk_token="<unk>", model_max_length=int(g["max_len"]))
    tok.save_pretrained(str(cdir))
    cfg = RobertaConfig(vocab_size=len(tok), hidden_size=512, num_hidden_layers=1, num_attention_heads=8, intermediate_size=128, max_position_embeddings=64,
                        type_vocab_size=1, pad_token_id=tok.pad_token_id, layer_norm_eps=1e-5)
    torch.manual_seed(8)
    hf = RobertaModel(cfg, add_pooling_layer=False)
    hf.save_pretrained(str(cdir))
    return str(udir / "mol_pre_no_h_220816.pt"), str(cdir), tower1, hf


def _tiny_roberta_ns(vocab):
    from types import SimpleNamespace
    return SimpleNamespace(layers=1, dim=512, heads=8, ffn=128, vocab=vocab, max_pos=64, type_vocab=1, pad_idx=1, ln_eps=1e-5, hidden_dropout=0.1, attn_dropout=0.1)


def _construct(unimol_dir, chemberta_dir, task="classification", **kw):
    from mmdti_hip.models import mm_model as mm
    out_dim = 1 if task == "regression" else 2
    # (exactly the keyword form NNMODEL_REGISTER['mm_model'](**params) receives, models/nnmodel.py:115-116)
    return mm.MM_Model(output_dim=out_dim, task=task, unimol_dir=unimol_dir, chemberta_dir=chemberta_dir, **kw)


def test_constructor_loads_dictionary_towers_and_tokenizer(tmp_path, golden):
    from mmdti_hip.models.bert_layers import RobertaTower
    from mmdti_hip.unicore_compat import Dictionary
    unimol_dir, chemberta_dir, tower1, hf = _write_inputs(tmp_path, golden)
    d = Dictionary.load(os.path.join(os.path.dirname(unimol_dir), "mol.dict.txt"))
    assert len(d) == len(MOL_SYMBOLS) and d.pad() == 0 and d.index("C") == 4
    model = _construct(unimol_dir, chemberta_dir)
    assert len(model.dictionary) == len(MOL_SYMBOLS) + 1 and model.mask_idx == len(MOL_SYMBOLS) and model.padding_idx == 0       # + [MASK]
    assert model.gbf.mul.weight.shape[0] == (len(MOL_SYMBOLS) + 1) ** 2
    # tower 1: every checkpoint tensor arrived (strict on the tower-1 keys, the unknown pre-training key ignored)
    sd = model.state_dict()
    for k, v in tower1.items():
        assert torch.equal(sd[k], v), k
    # tower 2: HF's own save_pretrained directory, read without the HF model classes
    hsd = hf.state_dict()
    n = 0
    for k, v in hsd.items():
        if "position_ids" in k or not v.is_floating_point():
            continue
        assert torch.equal(sd["bert." + k], v), k
        n += 1
    assert n >= 16 + 5
    assert isinstance(model.bert, RobertaTower) and model.bert.cfg.layers == 1 and model.bert.cfg.dim == 512 and model.bert.cfg.ln_eps == 1e-5
    # the tokenizer hand-off: batch_collate_fn tokenizes the SMILES of a reference-shaped sample list
    g = golden("g9_collate")
    samples = samples_from(g)
    batch, label = model.batch_collate_fn(samples)
    assert list(batch)[-2:] == ["input_ids", "attention_mask"] and batch["input_ids"].shape[0] == len(samples)
    assert torch.equal(batch["input_ids"], T(g["o_input_ids"])) if "o_input_ids" in g else True
    assert batch["src_tokens"].dtype == torch.int64 and batch["src_distance"].dtype == torch.float32 and label is not None
    # a checkpoint that lacks a tower-1 tensor must not load silently (silent weight-name mismatch)
    broken = tmp_path / "broken"
    broken.mkdir()
    u2, c2, _, _ = _write_inputs(broken, golden, drop_key="encoder.layers.3.fc1.weight")
    with pytest.raises(RuntimeError, match="tower-1"):
        _construct(u2, c2)
    # a RoBERTa directory that lacks a tensor must not load silently either
    from safetensors.torch import load_file, save_file
    st = os.path.join(c2, "model.safetensors")
    w = load_file(st)
    w.pop(next(k for k in w if "intermediate.dense.weight" in k))
    save_file(w, st)
    with pytest.raises(RuntimeError, match="lacks parameters"):
        RobertaTower.from_pretrained(c2)
    # the hidden size of tower 2 must be the fusion block's (mm_model.py: cross-modal hidden size 512)
    cfgp = os.path.join(chemberta_dir, "config.json")
    c = json.load(open(cfgp))
    c["hidden_act"] = "relu"
    json.dump(c, open(cfgp, "w"))
    with pytest.raises(NotImplementedError):
        RobertaTower.from_pretrained(chemberta_dir)


@pytest.mark.gpu
def test_constructed_model_trains_and_predicts_through_the_trainer(tmp_path, golden):
    """NNModel.run / evaluate (models/nnmodel.py:150-232) with the constructed model: Trainer.fit_predict for one epoch, the checkpoint on
    disk, then predict(load_model=True) into a FRESHLY constructed model -- which must reproduce the trained model's predictions."""
    from mmdti_hip.tasks import Trainer
    unimol_dir, chemberta_dir, tower1, _ = _write_inputs(tmp_path, golden)
    g = golden("g10_trainer_cls")
    train, valid = samples_from(g, "train_"), samples_from(g, "valid_")
    assert max(int(np.max(s[0]["src_tokens"])) for s in train + valid) < len(MOL_SYMBOLS)
    hp = dict(json.loads(str(g["hp_json"])), use_cuda=True, epochs=1)
    torch.manual_seed(99)
    model = _construct(unimol_dir, chemberta_dir)
    before = model.encoder.layers[7].fc1.weight.detach().clone()
    trainer = Trainer(save_path=str(tmp_path), **hp)
    act = lambda x: torch.softmax(x, dim=-1)[:, 1:]
    y_val = trainer.fit_predict(model, train, valid, None, act, str(tmp_path), 0, None, return_infonce_loss=True, return_ct_loss=True, use_weight=False)
    assert y_val.shape[0] == len(valid) and np.isfinite(y_val).all()
    steps = np.concatenate([h["steps"] for h in trainer.history])
    assert np.isfinite(steps).all() and steps.shape[0] == len(train) // hp["batch_size"]          # (one epoch; the training loader drops the last partial batch)
    assert not torch.equal(model.encoder.layers[7].fc1.weight.detach().cpu(), before)                 # it trained
    ck_path = os.path.join(str(tmp_path), "model_0.pth")
    ck = torch.load(ck_path, map_location="cpu", weights_only=True)["model_state_dict"]
    assert set(ck) == set(model.state_dict())
    # evaluate(): a fresh model from the same two paths, the checkpoint loaded by predict
    torch.manual_seed(5)
    fresh = _construct(unimol_dir, chemberta_dir)
    y_new, _, _ = Trainer(save_path=str(tmp_path), **hp).predict(fresh, valid, None, act, str(tmp_path), 0, None, epoch=1, load_model=True)
    y_old, _, _ = trainer.predict(model, valid, None, act, str(tmp_path), 0, None, epoch=1, load_model=True)
    assert np.allclose(y_new, y_old, rtol=0, atol=1e-6), float(np.abs(y_new - y_old).max())
    assert np.allclose(y_old, y_val, rtol=0, atol=2e-3)           # (fit_predict returns the BEST epoch's validation predictions: one epoch here)
