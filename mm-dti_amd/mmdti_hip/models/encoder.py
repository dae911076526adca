"""Working stand-in for the reference's ``models/encoder.py`` split-tower classes.

The reference file is dead code that cannot be imported anywhere (it imports the non-existent ``.test_module`` and
``.rnc_loss`` at :3,:20 and calls ``AutoTokenizer.from_pretrained`` on an absolute path at import time, :29).  The north
star nevertheless lists it as replaced, so this module offers the same two class names with the forward signatures of
:458-502 and :555-557, executing on the gfx950 kernels:

  ``UnimolEncoder(output_dim=2, **params).forward(src_tokens, src_distance, src_edge_type) -> [B, N, 512]``
  ``ChembertaEncoder(model_name_or_path, **params).forward(input_ids, attention_mask) -> [B, L, H]``

State-dict keys equal the corresponding sub-trees of ``MM_Model`` (``embed_tokens``, ``encoder``, ``gbf``, ``gbf_proj`` /
``bert``), so tower weights can be moved between the fused model and the stand-alone encoders.
"""
import os

import torch
import torch.nn as nn

from ..unicore_compat import Dictionary, init_bert_params
from ..functional import PairBiasFn, EmbeddingFn
from .. import ops
from .transformers import TransformerEncoderWithPair
from .bert_layers import RobertaTower
from ..collate import FIELD_RULES, collate_field, stack_labels, tokenize
from .mm_model import GaussianLayer, NonLinearHead, molecule_architecture, fds_config, crossmodal_config


class UnimolEncoder(nn.Module):
    def __init__(self, output_dim=2, **params):
        super().__init__()
        self.cross_cfg = crossmodal_config()
        self.fds_cfg = fds_config()
        self.args = params.get('_mol_args') or molecule_architecture()
        self.output_dim = output_dim
        self.data_type = 'molecule'
        self.remove_hs = params.get('remove_hs', False)
        self.use_fds = params.get('fds', False)
        dictionary = params.get('_dictionary')
        self.pretrain_path = params.get('unimol_dir', '')
        if dictionary is None:
            if not self.pretrain_path:
                dictionary = Dictionary.default_molecule()
            else:
                dictionary = Dictionary.load(os.path.join(os.path.dirname(self.pretrain_path), 'mol.dict.txt'))
        self.dictionary = dictionary
        self.mask_idx = self.dictionary.add_symbol("[MASK]", is_special=True)
        self.padding_idx = self.dictionary.pad()
        a = self.args
        self.embed_tokens = nn.Embedding(len(self.dictionary), a.encoder_embed_dim, self.padding_idx)
        self.encoder = TransformerEncoderWithPair(
            encoder_layers=a.encoder_layers, embed_dim=a.encoder_embed_dim, ffn_embed_dim=a.encoder_ffn_embed_dim,
            attention_heads=a.encoder_attention_heads, emb_dropout=a.emb_dropout, dropout=a.dropout,
            attention_dropout=a.attention_dropout, activation_dropout=a.activation_dropout, max_seq_len=a.max_seq_len,
            activation_fn=a.activation_fn, no_final_head_layer_norm=a.delta_pair_repr_norm_loss < 0)
        K = params.get('_gbf_K', 128)
        self.gbf_proj = NonLinearHead(K, a.encoder_attention_heads, a.activation_fn)
        self.gbf = GaussianLayer(K, len(self.dictionary) * len(self.dictionary))
        self.apply(init_bert_params)
        if self.pretrain_path:
            self.load_pretrained_weights(self.pretrain_path)

    def load_pretrained_weights(self, path):
        sd = torch.load(path, map_location="cpu", weights_only=True)
        sd = sd['model'] if 'model' in sd else sd
        own = self.state_dict()
        missing = [k for k in own if k not in sd]
        if missing:
            raise RuntimeError(f"Uni-Mol checkpoint {path} lacks {len(missing)} parameters, e.g. {missing[:5]}")
        self.load_state_dict({k: v for k, v in sd.items() if k in own}, strict=True)

    def forward(self, src_tokens, src_distance, src_edge_type):
        padding_mask = src_tokens.eq(self.padding_idx)
        x = EmbeddingFn.apply(self.embed_tokens.weight, src_tokens, self.padding_idx)
        N = src_distance.shape[-1]
        bias = PairBiasFn.apply(self.gbf.means.weight, src_distance.float(), src_edge_type, self.gbf, self.gbf_proj, ops.pair_ld(N))
        encoder_rep, _, _ = self.encoder.encode(x, bias, padding_mask)
        return encoder_rep

    def batch_collate_fn(self, samples):
        """The 3D-conformer fields of MM_Model.batch_collate_fn (layout rules: mmdti_hip.collate); other keys are skipped."""
        feats = [s[0] for s in samples]
        batch = {}
        for key in feats[0]:
            if key in FIELD_RULES:
                batch[key] = collate_field(key, (f[key] for f in feats), self.padding_idx)
        return batch, stack_labels(samples)


class ChembertaEncoder(nn.Module):
    def __init__(self, model_name_or_path=None, **params):
        super().__init__()
        cfg = params.get('_roberta_cfg')
        if cfg is not None:
            self.bert = RobertaTower(cfg)
            self.bert.apply(init_bert_params)
            self.tokenizer = params.get('_tokenizer')
        else:
            if not model_name_or_path:
                raise ValueError("ChembertaEncoder needs a local HuggingFace RoBERTa directory (or _roberta_cfg for random init)")
            self.bert = RobertaTower.from_pretrained(model_name_or_path)
            from transformers import AutoTokenizer
            self.tokenizer = AutoTokenizer.from_pretrained(model_name_or_path)

    def forward(self, input_ids, attention_mask):
        return self.bert(input_ids=input_ids, attention_mask=attention_mask, return_dict=True)[0]

    def batch_collate_fn(self, samples):
        """The SMILES half of the collate (no truncation here: encoder.py:563 passes padding=True only)."""
        feats = [s[0] for s in samples]
        batch = {}
        if 'smile' in feats[0]:
            batch['input_ids'], batch['attention_mask'] = tokenize(self.tokenizer, (f['smile'] for f in feats), truncation=False)
        return batch, stack_labels(samples)
