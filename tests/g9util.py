"""Shared helpers of the G9/G10 tests (fixtures generated from the reference's own run: tests/golden/make_golden_g9.py)."""
from types import SimpleNamespace

import numpy as np
import torch

from oracle import mmdti_oracle as O


def T(a):
    return torch.from_numpy(np.asarray(a))


def samples_from(g, prefix=""):
    """DataHub-shaped samples [(feature dict, label)] stored flat in a fixture."""
    n = int(g[prefix + "n_samples"])
    smiles = [str(s) for s in g[prefix + "smiles"]]
    out = []
    for i in range(n):
        d = {k: g[f"{prefix}s{i}_{k}"] for k in ("src_tokens", "src_distance", "src_coord", "src_edge_type")}
        d["smile"] = smiles[i]
        if f"{prefix}s{i}_weights" in g:
            d["weights"] = float(g[f"{prefix}s{i}_weights"])
        out.append((d, g[f"{prefix}s{i}_label"]))
    return out


def zero_dropout_cfg(u: dict, r: dict, c: dict, task, out_dim):
    return O.ModelCfg(unimol=O.UniMolCfg(emb_dropout=0.0, dropout=0.0, attn_dropout=0.0, pooler_dropout=0.0, **u),
                      roberta=O.RobertaCfg(hidden_dropout=0.0, attn_dropout=0.0, **r),
                      cross=O.CrossCfg(hidden_dropout=0.0, attn_dropout=0.0, **c), task=task, output_dim=out_dim, infonce_dropout=0.0)


TINY_U = dict(layers=2, dim=64, ffn=128, heads=8, K=128, vocab=31)
TINY_C = dict(dim=64, heads=4, ffn=128)
REF_U = dict(layers=15, dim=512, ffn=2048, heads=64, K=128, vocab=31)
REF_C = dict(dim=512, heads=16, ffn=2048)


def tiny_cfg(task, vocab_rob):
    return zero_dropout_cfg(TINY_U, dict(layers=2, dim=64, heads=4, ffn=128, vocab=vocab_rob, max_pos=40), TINY_C, task,
                            1 if task == "regression" else 2)


def refarch_cfg(task, vocab_rob):
    return zero_dropout_cfg(REF_U, dict(layers=6, dim=512, heads=8, ffn=2048, vocab=vocab_rob, max_pos=514), REF_C, task,
                            1 if task == "regression" else 2)


def tokenizer_from(tok_json, max_len):
    from tokenizers import Tokenizer
    from transformers import PreTrainedTokenizerFast
    return PreTrainedTokenizerFast(tokenizer_object=Tokenizer.from_str(tok_json), bos_token="<s>", eos_token="</s>",
                                   pad_token="<pad>", unk_token="<unk>", model_max_length=max_len)


def product_model(ocfg: O.ModelCfg, tokenizer=None, dropout=False, **params):
    """The product's MM_Model at the architecture of an oracle config (dropout probabilities 0 unless ``dropout``)."""
    from mmdti_hip.models import mm_model as mm
    u, r, c = ocfg.unimol, ocfg.roberta, ocfg.cross
    mol = mm.molecule_architecture()
    mol.encoder_layers, mol.encoder_embed_dim, mol.encoder_ffn_embed_dim, mol.encoder_attention_heads = u.layers, u.dim, u.ffn, u.heads
    cross = mm.crossmodal_config()
    cross.hidden_size, cross.num_attention_heads, cross.intermediate_size = c.dim, c.heads, c.ffn
    rcfg = SimpleNamespace(layers=r.layers, dim=r.dim, heads=r.heads, ffn=r.ffn, vocab=r.vocab, max_pos=r.max_pos, type_vocab=1, pad_idx=1,
                           ln_eps=1e-12, hidden_dropout=0.1, attn_dropout=0.1)
    if not dropout:
        mol.dropout = mol.emb_dropout = mol.attention_dropout = mol.pooler_dropout = 0.0
        cross.hidden_dropout_prob = cross.attention_probs_dropout_prob = 0.0
        rcfg.hidden_dropout = rcfg.attn_dropout = 0.0
    model = mm.MM_Model.from_configs(ocfg.output_dim, ocfg.task, mol_args=mol, roberta_cfg=rcfg, cross_cfg=cross, gbf_K=u.K,
                                     _tokenizer=tokenizer, **params)
    if not dropout:
        model.infonce.embed_dropout = 0.0
    return model


def load_fixture_weights(model, sd):
    """strict load of a reference state dict (floating tensors), tolerating only keys the product legitimately lacks /
    adds (HF's integer position-id buffers; FDS buffers handled by the caller)."""
    sd = {k: v for k, v in sd.items() if torch.is_tensor(v) and v.is_floating_point()}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    missing = [k for k in missing if not k.startswith(("FDS.", "bert.pooler."))]       # (the oracle's parameter set has no pooler: it gets no gradient)
    assert not missing and not unexpected, (missing[:5], unexpected[:5])


def rel_l2(a, b):
    a = a.detach().double().cpu().flatten() if torch.is_tensor(a) else T(a).double().flatten()
    b = b.detach().double().cpu().flatten() if torch.is_tensor(b) else T(b).double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def cosine(a, b):
    a = a.detach().double().cpu().flatten() if torch.is_tensor(a) else T(a).double().flatten()
    b = b.detach().double().cpu().flatten() if torch.is_tensor(b) else T(b).double().flatten()
    return float((a @ b) / (a.norm() * b.norm() + 1e-30))


def host_fields(batch_cpu, pad_idx=0):
    """The host-side descriptors collate.device_payload attaches to a collated batch (atom / token counts, packable): with them
    MM_Model runs a ragged batch on packed token rows (mmdti_hip/packing.py)."""
    from mmdti_hip.collate import device_payload, HOST_FIELDS
    full = device_payload({k: v for k, v in batch_cpu.items()}, pad_idx=pad_idx)
    return {k: full[k] for k in HOST_FIELDS if k in full}


_BANDS = {}


def record_band(name, **vals):
    """Measured values behind a tolerance band, written to gpurun_out/grad_bands.json on the GPU box: every gradient band of the GPU
    suite states `measured x 1.3` next to the assert, and this file is where `measured` comes from (copied to profiles/ per round)."""
    import json, os
    _BANDS[name] = {k: (float(v) if not isinstance(v, str) else v) for k, v in vals.items()}
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "grad_bands.json"), "w") as f:
        json.dump(_BANDS, f, indent=1, sort_keys=True)
