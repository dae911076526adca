"""Drop-in for the reference's ``models/transformers.py`` (TransformerEncoderWithPair), MI355X-native.

Same class name, constructor keywords, forward signature, 5-tuple return and parameter names as
/root/reference/models/transformers.py:14-183 (and, through ``unicore_compat``, Uni-Core's checkpoint keys
``emb_layer_norm.*``, ``layers.{i}.self_attn.in_proj.*`` ...).  The arithmetic -- LayerNorms, the four Linears per
layer, the pair-bias attention with S chained through the 15 layers -- runs in hand-written gfx950 kernels through
``functional.PairEncoderFn`` (forward and backward).
"""
from typing import Optional

import torch
import torch.nn as nn

from ..unicore_compat import TransformerEncoderLayer, LayerNorm
from ..functional import PairEncoderFn


class TransformerEncoderWithPair(nn.Module):
    def __init__(
        self,
        encoder_layers: int = 6,
        embed_dim: int = 768,
        ffn_embed_dim: int = 3072,
        attention_heads: int = 8,
        emb_dropout: float = 0.1,
        dropout: float = 0.1,
        attention_dropout: float = 0.1,
        activation_dropout: float = 0.0,
        max_seq_len: int = 256,
        activation_fn: str = "gelu",
        post_ln: bool = False,
        no_final_head_layer_norm: bool = False,
    ) -> None:
        super().__init__()
        if embed_dim // attention_heads != 8 or embed_dim % attention_heads:
            raise ValueError("the gfx950 pair-attention kernel is specialised for head_dim 8 "
                             f"(embed_dim={embed_dim}, attention_heads={attention_heads})")
        if activation_dropout != 0.0:
            raise NotImplementedError("activation_dropout != 0 is not on the MM-DTI path (mm_model.py:333)")
        self.emb_dropout = emb_dropout
        self.max_seq_len = max_seq_len
        self.embed_dim = embed_dim
        self.attention_heads = attention_heads
        self.dropout = dropout
        self.attention_dropout = attention_dropout
        self.emb_layer_norm = LayerNorm(self.embed_dim)
        self.final_layer_norm = LayerNorm(self.embed_dim) if not post_ln else None
        self.final_head_layer_norm = LayerNorm(attention_heads) if not no_final_head_layer_norm else None
        self.layers = nn.ModuleList(
            [
                TransformerEncoderLayer(
                    embed_dim=self.embed_dim,
                    ffn_embed_dim=ffn_embed_dim,
                    attention_heads=attention_heads,
                    dropout=dropout,
                    attention_dropout=attention_dropout,
                    activation_dropout=activation_dropout,
                    activation_fn=activation_fn,
                    post_ln=post_ln,
                )
                for _ in range(encoder_layers)
            ]
        )

    def encode(self, emb: torch.Tensor, pair_bias: torch.Tensor, padding_mask: Optional[torch.Tensor], key_tiles: Optional[torch.Tensor] = None,
               pack=None):
        """Fast path used by MM_Model: pair_bias is [B,H,N,ld] fp32 (internal layout).  -> (x [B,N,D], S_last, x_pre).
        key_tiles ([B] int32 on the device, tiled layout only): 16-key tiles of each molecule that hold a real key -- the pair
        attention kernels skip the all-padding key tiles past it (ragged batches).
        pack (packing.PackedRows, with key_tiles): emb / padding_mask / x are packed rows [pack.M, ...] -- real tokens plus one
        representative pad row per molecule (see packing.py)."""
        return PairEncoderFn.apply(emb, pair_bias, padding_mask, self, self.training, key_tiles, pack)

    def forward(
        self,
        emb: torch.Tensor,
        attn_mask: Optional[torch.Tensor] = None,
        padding_mask: Optional[torch.Tensor] = None,
    ):
        """Reference signature and 5-tuple (models/transformers.py:96-183).  All five returns are differentiable, as in the
        reference: the encoder output through the kernels' own backward; the four auxiliary ones (discarded by MM_Model,
        mm_model.py:559) as plain tensor glue on the encoder's last logits and its pre-final-LN stream, whose gradients enter the
        same kernel backward (``PairEncoderFn(aux_grads=True)``)."""
        assert attn_mask is not None
        bsz, seq_len = emb.size(0), emb.size(1)
        H = self.attention_heads
        bias = attn_mask.view(bsz, H, seq_len, seq_len)
        pad = None
        if padding_mask is not None:
            # the reference merges the key-padding mask into the CALLER's tensor in place (:122-135)
            pad = padding_mask.unsqueeze(1).unsqueeze(2).to(torch.bool)
            bias.masked_fill_(pad, float("-inf"))
        x, s_last, x_pre = PairEncoderFn.apply(emb.float(), bias.float(), padding_mask, self, self.training, None, None, True)
        token_mask = 1.0 - padding_mask.float() if padding_mask is not None else torch.ones(bsz, seq_len, device=emb.device)
        delta = s_last - bias.float()                  # (-inf - -inf at padded keys: filled next, as :163-164 does)
        if pad is not None:
            delta = delta.masked_fill(pad, 0.0)
        attn = s_last.permute(0, 2, 3, 1).contiguous()
        delta = delta.permute(0, 2, 3, 1).contiguous()

        def norm_loss(t, eps=1e-10, tolerance=1.0):
            max_norm = t.shape[-1] ** 0.5
            norm = torch.sqrt(torch.sum(t.float() ** 2, dim=-1) + eps)
            return torch.nn.functional.relu((norm - max_norm).abs() - tolerance)

        pair_mask = token_mask[..., None] * token_mask[..., None, :]
        dn = norm_loss(delta)
        delta_norm = (torch.sum(pair_mask * dn, dim=(-1, -2)) / (1e-10 + torch.sum(pair_mask, dim=(-1, -2)))).mean()
        xn = norm_loss(x_pre)
        x_norm = (torch.sum(token_mask * xn, dim=-1) / (1e-10 + torch.sum(token_mask, dim=-1))).mean()
        if self.final_head_layer_norm is not None:
            delta = self.final_head_layer_norm(delta)
        return x, attn, delta, x_norm, delta_norm
