"""The sliver of Uni-Core's surface that the reference's hot path touches, as parameter containers
with Uni-Core's parameter names.  No arithmetic lives here: forward passes run through ``mmdti_hip.functional``.

Uni-Core (dptech-corp/Uni-Core, unpinned by the reference) is absent from the reference tree and from this image; names
and semantics below follow its published modules as used at models/transformers.py:11,69-91 and
models/mm_model.py:13-16,435-441,472.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F


class LayerNorm(nn.Module):
    """unicore.modules.LayerNorm: weight/bias/eps holder (eps 1e-5)."""

    def __init__(self, normalized_shape, eps=1e-5, elementwise_affine=True):
        super().__init__()
        if isinstance(normalized_shape, int):
            normalized_shape = (normalized_shape,)
        self.normalized_shape = tuple(normalized_shape)
        self.eps = eps
        assert elementwise_affine
        self.weight = nn.Parameter(torch.ones(*self.normalized_shape))
        self.bias = nn.Parameter(torch.zeros(*self.normalized_shape))

    def forward(self, x):
        from .functional_small import layer_norm_autograd
        return layer_norm_autograd(x, self.weight, self.bias, self.eps)


class SelfMultiheadAttention(nn.Module):
    """Parameter container with Uni-Core's names: in_proj [3D,D], out_proj [D,D]."""

    def __init__(self, embed_dim, num_heads, dropout=0.1, bias=True, scaling_factor=1):
        super().__init__()
        self.embed_dim, self.num_heads, self.dropout = embed_dim, num_heads, dropout
        self.head_dim = embed_dim // num_heads
        assert self.head_dim * num_heads == embed_dim, "embed_dim must be divisible by num_heads"
        self.scaling = (self.head_dim * scaling_factor) ** -0.5
        self.in_proj = nn.Linear(embed_dim, embed_dim * 3, bias=bias)
        self.out_proj = nn.Linear(embed_dim, embed_dim, bias=bias)


class TransformerEncoderLayer(nn.Module):
    """Parameter container of unicore.modules.TransformerEncoderLayer (pre-LN)."""

    def __init__(self, embed_dim=768, ffn_embed_dim=3072, attention_heads=8, dropout=0.1, attention_dropout=0.1,
                 activation_dropout=0.0, activation_fn="gelu", post_ln=False):
        super().__init__()
        if post_ln:
            raise NotImplementedError("post_ln=True is not on the MM-DTI path (molecule_architecture: post_ln=False)")
        if activation_fn != "gelu":
            raise NotImplementedError("only activation_fn='gelu' is on the MM-DTI path")
        self.embed_dim, self.attention_heads = embed_dim, attention_heads
        self.dropout, self.attention_dropout, self.activation_dropout = dropout, attention_dropout, activation_dropout
        self.self_attn = SelfMultiheadAttention(embed_dim, attention_heads, dropout=attention_dropout)
        self.self_attn_layer_norm = LayerNorm(embed_dim)
        self.fc1 = nn.Linear(embed_dim, ffn_embed_dim)
        self.fc2 = nn.Linear(ffn_embed_dim, embed_dim)
        self.final_layer_norm = LayerNorm(embed_dim)


def get_activation_fn(name):
    """unicore.utils.get_activation_fn (names only; used for config validation)."""
    table = {"gelu": F.gelu, "tanh": torch.tanh, "relu": F.relu, "linear": lambda x: x}
    if name not in table:
        raise RuntimeError(f"--activation-fn {name} not supported")
    return table[name]


def init_bert_params(module):
    """unicore.modules.init_bert_params: N(0,0.02) weights, zero biases, zeroed padding rows."""
    if isinstance(module, nn.Linear):
        module.weight.data.normal_(mean=0.0, std=0.02)
        if module.bias is not None:
            module.bias.data.zero_()
    if isinstance(module, nn.Embedding):
        module.weight.data.normal_(mean=0.0, std=0.02)
        if module.padding_idx is not None:
            module.weight.data[module.padding_idx].zero_()


class Dictionary:
    """unicore.data.Dictionary: one symbol per line (optional count column), index = line order."""

    def __init__(self):
        self.symbols, self.indices, self.specials = [], {}, set()
        self.bos_word, self.pad_word, self.eos_word, self.unk_word = "[CLS]", "[PAD]", "[SEP]", "[UNK]"

    def __len__(self):
        return len(self.symbols)

    def __contains__(self, sym):
        return sym in self.indices

    def index(self, sym):
        return self.indices.get(sym, self.indices.get(self.unk_word))

    def add_symbol(self, word, is_special=False):
        if is_special:
            self.specials.add(word)
        if word in self.indices:
            return self.indices[word]
        idx = len(self.symbols)
        self.indices[word] = idx
        self.symbols.append(word)
        return idx

    def bos(self):
        return self.index(self.bos_word)

    def pad(self):
        return self.index(self.pad_word)

    def eos(self):
        return self.index(self.eos_word)

    def unk(self):
        return self.index(self.unk_word)

    @classmethod
    def load(cls, path):
        d = cls()
        with open(path, "r", encoding="utf-8") as f:
            for line in f:
                parts = line.rstrip().rsplit(" ", 1)
                if parts and parts[0]:
                    d.add_symbol(parts[0])
        return d

    # Uni-Mol's mol.dict.txt as recalled in SURVEY.md 8c; used for synthetic data only.
    DEFAULT_MOL_SYMBOLS = ("[PAD] [CLS] [SEP] [UNK] C N O S H Cl F Br I Si P B Na K Al Ca Sn As Hg Fe Zn Cr Se Gd Au Li").split()

    @classmethod
    def default_molecule(cls):
        d = cls()
        for s in cls.DEFAULT_MOL_SYMBOLS:
            d.add_symbol(s)
        return d
