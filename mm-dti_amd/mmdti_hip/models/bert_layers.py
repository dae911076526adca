"""Parameter containers with the sub-module names shared by HF ``RobertaLayer`` (tower 2, mm_model.py:475,562) and the
reference's ``BertCrossAttentionLayer`` (models/mm_module.py:470-626): ``attention.self.{query,key,value}``,
``attention.output.{dense,LayerNorm}``, ``intermediate.dense``, ``output.{dense,LayerNorm}``.  No arithmetic here -- the
layers execute through ``functional._bert_layer_fwd/_bwd``."""
import json
import os
from types import SimpleNamespace

import torch
import torch.nn as nn

from ..unicore_compat import LayerNorm
from ..functional import RobertaEncoderFn, CrossLayerFn


class _SelfAttnParams(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.query = nn.Linear(dim, dim)
        self.key = nn.Linear(dim, dim)
        self.value = nn.Linear(dim, dim)


class _SelfOutputParams(nn.Module):
    def __init__(self, in_dim, dim, eps):
        super().__init__()
        self.dense = nn.Linear(in_dim, dim)
        self.LayerNorm = LayerNorm(dim, eps=eps)


class _AttentionParams(nn.Module):
    def __init__(self, dim, eps):
        super().__init__()
        self.self = _SelfAttnParams(dim)
        self.output = _SelfOutputParams(dim, dim, eps)


class _IntermediateParams(nn.Module):
    def __init__(self, dim, ffn):
        super().__init__()
        self.dense = nn.Linear(dim, ffn)


class BertLayerParams(nn.Module):
    def __init__(self, dim, ffn, eps):
        super().__init__()
        self.attention = _AttentionParams(dim, eps)
        self.intermediate = _IntermediateParams(dim, ffn)
        self.output = _SelfOutputParams(ffn, dim, eps)


class BertCrossEncoder(nn.Module):
    """models/mm_module.py:663-677: ``layer_num`` cross-attention layers; forward returns the list of layer outputs."""

    def __init__(self, config, layer_num):
        super().__init__()
        if config.hidden_act != "gelu":
            raise NotImplementedError("only hidden_act='gelu' is on the MM-DTI path")
        self.cfg = SimpleNamespace(heads=config.num_attention_heads, ln_eps=config.layer_norm_eps,
                                   hidden_dropout=config.hidden_dropout_prob, attn_dropout=config.attention_probs_dropout_prob)
        self.layer = nn.ModuleList([BertLayerParams(config.hidden_size, config.intermediate_size, config.layer_norm_eps)
                                    for _ in range(layer_num)])

    def forward(self, s1_hidden_states, s2_hidden_states, s2_attention_mask, output_all_encoded_layers=True, packs=None):
        """packs = (PackedRows of s1, PackedRows of s2): both inputs are packed rows [M, D] and the mask is implied (the keys of
        a sequence are the real rows of s2) -- see packing.py."""
        if packs is None:
            B, Lk = s2_hidden_states.shape[0], s2_hidden_states.shape[1]
            key_add = s2_attention_mask.reshape(B, Lk).float()
        else:
            key_add = None
        outs = []
        for layer in self.layer:
            s1_hidden_states = CrossLayerFn.apply(s1_hidden_states.float(), s2_hidden_states.float(), key_add, layer, self.cfg, self.training, packs)
            if output_all_encoded_layers:
                outs.append(s1_hidden_states)
        if not output_all_encoded_layers:
            outs.append(s1_hidden_states)
        return outs


class _RobertaEmbeddings(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.word_embeddings = nn.Embedding(cfg.vocab, cfg.dim, padding_idx=cfg.pad_idx)
        self.position_embeddings = nn.Embedding(cfg.max_pos, cfg.dim, padding_idx=cfg.pad_idx)
        self.token_type_embeddings = nn.Embedding(cfg.type_vocab, cfg.dim)
        self.LayerNorm = LayerNorm(cfg.dim, eps=cfg.ln_eps)


class _RobertaEncoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.layer = nn.ModuleList([BertLayerParams(cfg.dim, cfg.ffn, cfg.ln_eps) for _ in range(cfg.layers)])


class _Pooler(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.dense = nn.Linear(dim, dim)


class RobertaTower(nn.Module):
    """HF ``RobertaModel`` stand-in with identical state-dict keys (``embeddings.*``, ``encoder.layer.{i}.*``,
    ``pooler.dense.*``); ``forward(input_ids, attention_mask, return_dict=True)[0]`` is the last hidden state
    (mm_model.py:562).  The pooler parameters exist for checkpoint compatibility and, as in the reference's usage,
    receive no gradient."""

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.embeddings = _RobertaEmbeddings(cfg)
        self.encoder = _RobertaEncoder(cfg)
        self.pooler = _Pooler(cfg.dim)
        for p in self.pooler.parameters():
            p.requires_grad_(False)

    # attribute view used by RobertaEncoderFn
    @property
    def word(self):
        return self.embeddings.word_embeddings.weight

    @property
    def position(self):
        return self.embeddings.position_embeddings.weight

    @property
    def token_type(self):
        return self.embeddings.token_type_embeddings.weight

    @property
    def emb_ln_w(self):
        return self.embeddings.LayerNorm.weight

    @property
    def emb_ln_b(self):
        return self.embeddings.LayerNorm.bias

    @property
    def layers(self):
        return self.encoder.layer

    def forward(self, input_ids, attention_mask=None, return_dict=True, pack=None, **kwargs):
        """pack (packing.PackedRows of the right-padded input_ids, lengths = attention_mask.sum(1), masked slots holding the pad
        id): the last hidden state comes back as packed rows [pack.M, D] instead of [B, L, D]."""
        if attention_mask is None:
            attention_mask = torch.ones_like(input_ids)
        if input_ids.shape[1] + self.cfg.pad_idx + 1 > self.cfg.max_pos:
            raise ValueError(f"sequence length {input_ids.shape[1]} exceeds max_position_embeddings {self.cfg.max_pos}")
        out = RobertaEncoderFn.apply(self.word, input_ids, attention_mask, self, self.training, pack)
        return (out,)

    @classmethod
    def config_from_hf_json(cls, path):
        """Read the fields the tower needs from a HuggingFace ``config.json`` (mm_model.py:475 from_pretrained dir)."""
        with open(path) as f:
            c = json.load(f)
        if c.get("hidden_act", "gelu") != "gelu":
            raise NotImplementedError("only hidden_act='gelu' RoBERTa checkpoints are supported")
        if c.get("position_embedding_type", "absolute") != "absolute":
            raise NotImplementedError("only absolute position embeddings are supported")
        return SimpleNamespace(layers=c["num_hidden_layers"], dim=c["hidden_size"], heads=c["num_attention_heads"],
                               ffn=c["intermediate_size"], vocab=c["vocab_size"], max_pos=c["max_position_embeddings"],
                               type_vocab=c.get("type_vocab_size", 1), pad_idx=c.get("pad_token_id", 1),
                               ln_eps=c.get("layer_norm_eps", 1e-12), hidden_dropout=c.get("hidden_dropout_prob", 0.1),
                               attn_dropout=c.get("attention_probs_dropout_prob", 0.1))

    @classmethod
    def from_pretrained(cls, directory):
        """Load a local HF RoBERTa directory (config.json + model.safetensors | pytorch_model.bin) without the HF stack."""
        cfg = cls.config_from_hf_json(os.path.join(directory, "config.json"))
        model = cls(cfg)
        st_path = os.path.join(directory, "model.safetensors")
        if os.path.exists(st_path):
            from safetensors.torch import load_file
            sd = load_file(st_path)
        else:
            sd = torch.load(os.path.join(directory, "pytorch_model.bin"), map_location="cpu", weights_only=True)
        sd = {k[len("roberta."):] if k.startswith("roberta.") else k: v for k, v in sd.items()}
        sd = {k: v for k, v in sd.items() if not k.startswith(("lm_head", "embeddings.position_ids"))}
        missing, unexpected = model.load_state_dict(sd, strict=False)
        missing = [k for k in missing if not k.startswith("pooler.")]
        if missing:
            raise RuntimeError(f"RoBERTa checkpoint in {directory} lacks parameters: {missing[:8]} ...")
        return model
