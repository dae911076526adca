"""CPU oracle for the MM-DTI dual-encoder contrastive fine-tune step.

TEST INFRASTRUCTURE ONLY.  This module is a plain fp32 PyTorch-CPU restatement
of the reference's algorithm for the hot path named in BASELINE.json.  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import it, and only as the checker -- the product path
(``mm-dti_amd/mmdti_hip``) never imports anything under ``oracle/`` and fails
loudly when the HIP library is missing.

Parity pinning.  The reference ships no tests, no golden
vectors and no fixtures.  The pieces of it that import in the build container
were run there and their outputs frozen under ``tests/golden/`` by
``tests/golden/make_golden.py``:
  * ``models/infonce.py``      -> info_nce / InfoNCE          (pinned)
  * ``models/contrastive.py``  -> CT_Regress/CT_Single/CT_Multi (pinned)
  * ``models/fds.py`` + ``utils/util.py`` -> FDS trajectory, calibrate_mean_var,
    pad_* helpers, kernel windows                              (pinned)
  * ``models/mm_module.py``    -> BertCrossEncoder             (pinned)
  * HuggingFace ``RobertaModel`` 5.15.0 (tower 2 arithmetic)   (pinned)
The Uni-Mol tower's arithmetic lives in Uni-Core, which is absent from the
reference tree and from this image (unpinned dependency): for
``unimol_layer`` / ``unimol_encoder`` and the Gaussian basis (whose file cannot
import without Uni-Core) this oracle is a restatement of the published
algorithm anchored on the reference's call sites -- **parity unpinned** for
those functions.

All tensors are fp32 unless stated.  ``emulate_bf16=True`` rounds the operands
of every contraction to bf16 at exactly the points where the HIP path stores
bf16 (see DESIGN.md "precision contract"), so kernel logic can be checked to
accumulation-order tolerance; with it off this is the pure-fp32 reference
arithmetic.

Parameter naming follows the reference's ``MM_Model.state_dict()`` keys
(SURVEY.md Appendix A), passed as a flat ``dict[str, Tensor]``.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
Params = Dict[str, Tensor]


# --------------------------------------------------------------------------
# configuration
# --------------------------------------------------------------------------
@dataclass
class UniMolCfg:
    """models/mm_model.py:325-343 (molecule_architecture)."""
    layers: int = 15
    dim: int = 512
    ffn: int = 2048
    heads: int = 64
    K: int = 128            # Gaussian kernels, mm_model.py:455
    vocab: int = 31         # len(dictionary) incl. [MASK]
    pad_idx: int = 0
    emb_dropout: float = 0.1
    dropout: float = 0.1
    attn_dropout: float = 0.1
    act_dropout: float = 0.0
    pooler_dropout: float = 0.2
    ln_eps: float = 1e-5


@dataclass
class RobertaCfg:
    """Tower 2 (HF RobertaModel).  [ASSUMED] defaults; real
    runs read config.json."""
    layers: int = 6
    dim: int = 512
    heads: int = 8
    ffn: int = 2048
    vocab: int = 600
    max_pos: int = 514
    type_vocab: int = 1
    pad_idx: int = 1
    ln_eps: float = 1e-12
    hidden_dropout: float = 0.1
    attn_dropout: float = 0.1


@dataclass
class CrossCfg:
    """models/mm_model.py:362-377 (crossmodal_config)."""
    dim: int = 512
    heads: int = 16
    ffn: int = 2048
    ln_eps: float = 1e-12
    hidden_dropout: float = 0.3
    attn_dropout: float = 0.2


@dataclass
class ModelCfg:
    unimol: UniMolCfg = field(default_factory=UniMolCfg)
    roberta: RobertaCfg = field(default_factory=RobertaCfg)
    cross: CrossCfg = field(default_factory=CrossCfg)
    task: str = "classification"     # 'classification' | 'regression' | 'multilabel_classification'
    output_dim: int = 2
    ct_w: float = 0.2
    infonce_dim: int = 50            # models/infonce.py:14
    infonce_temp: float = 0.1
    infonce_dropout: float = 0.1


# Rounding sites of the bf16 contract (tests/test_rounding_budget_cpu.py switches them one at a time to attribute the
# distance between the bf16 path and the fp32 reference): "w" GEMM weights, "x" GEMM inputs (LayerNorm / attention /
# GELU outputs), "qkv" the stored q|k|v of tower 1, "qkv2" q,k,v of towers 2 / fusion, "p" their attention probabilities,
# "proj" per-token InfoNCE projections (only when the head projects before pooling).
ALL_SITES = frozenset({"w", "x", "qkv", "qkv2", "p", "proj", "s16"})
# "s16": the pair logits of tower 1 (gbf bias, then every layer's S) are carried as fp16, rounded once per layer, and each
# layer's softmax runs on the rounded value -- the HIP path's compact pair planes, and what the reference's AMP path does
# (autocast makes attn_weights fp16).  Cost at the reference depth: profiles/r02_s16_budget_cpu.json.
BF16_SITES = set(ALL_SITES)
# storage type the sites round to: bfloat16 (the HIP path's contract) or float16 (what the reference's own AMP run uses,
# tasks/trainer.py:181-182 -- scratch/rounding_sites_fp16.py measures what that would buy; values saturate at 65504)
ROUND_DTYPE = torch.bfloat16
# The HIP path's DEFAULT contract since round 4 (ops.FWD_F16): the operands of FORWARD GEMMs -- sites "w", "x" and tower 1's stored
# q | k | v -- round to float16 (saturating), everything else ("qkv2", "p", "proj", the pair-bias block's own two Linears) stays
# bfloat16; the BACKWARD keeps bf16 operands, so the gradient that flows back through an fp16 site is rounded to bf16, exactly as it
# is through a bf16 site (the device converts a saved fp16 activation to bf16 inside the kernel that reads it; the 2^-9 that costs the
# weight gradients is below what these emulations resolve).  set_forward_fp16(False): every site bf16, the round-1..3 contract.
FWD_F16 = True
F16_SITES = frozenset({"w", "x", "qkv"})


def set_forward_fp16(on: bool):
    global FWD_F16
    FWD_F16 = bool(on)


class _RoundF16FwdBf16Bwd(torch.autograd.Function):
    """An fp16 forward-operand site: the value rounds to fp16 (saturating), its gradient to bf16."""

    @staticmethod
    def forward(ctx, x):
        return torch.clamp(x, min=-65504.0, max=65504.0).to(torch.float16).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(torch.float32)


class _RoundF16(torch.autograd.Function):
    """fp16 storage of a value (round to nearest even, saturating at the largest finite fp16; -inf stays -inf) whose
    gradient is NOT stored as fp16: a plain ``x.half().float()`` would send the gradient back through an fp16 cast and
    flush the small ones -- the HIP path keeps that chain in fp32."""

    @staticmethod
    def forward(ctx, x):
        return torch.clamp(x, max=65504.0).to(torch.float16).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        return g


def _r(x: Tensor, on: bool, site: str = "x", f16ok: bool = True) -> Tensor:
    """16-bit round-trip used by the ``emulate_bf16`` contract (f16ok=False: a site that stays bf16 in every mode)."""
    if not (on and site in BF16_SITES):
        return x
    if ROUND_DTYPE is torch.float16:
        return torch.clamp(x, min=-65504.0, max=65504.0).to(torch.float16).to(torch.float32)
    if FWD_F16 and f16ok and site in F16_SITES:
        return _RoundF16FwdBf16Bwd.apply(x)
    return x.to(ROUND_DTYPE).to(torch.float32)


def linear(x: Tensor, w: Tensor, b: Optional[Tensor], bf16: bool = False, f16ok: bool = True) -> Tensor:
    """nn.Linear.  Under the 16-bit contract both operands are rounded, the
    accumulation and bias add stay fp32."""
    return F.linear(_r(x, bf16, "x", f16ok), _r(w, bf16, "w", f16ok), b)


def gelu(x: Tensor) -> Tensor:
    """erf GELU (unicore get_activation_fn('gelu') == F.gelu; mm_module.py:204-210)."""
    return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


def layer_norm(x: Tensor, w: Tensor, b: Tensor, eps: float) -> Tensor:
    """Biased-variance LayerNorm, eps inside the sqrt (unicore LayerNorm ==
    F.layer_norm; BertLayerNorm mm_module.py:320-333; HF nn.LayerNorm)."""
    u = x.mean(-1, keepdim=True)
    s = (x - u).pow(2).mean(-1, keepdim=True)
    return (x - u) / torch.sqrt(s + eps) * w + b


def dropout(x: Tensor, p: float, training: bool) -> Tensor:
    return F.dropout(x, p=p, training=training) if (training and p > 0) else x


# --------------------------------------------------------------------------
# a0: batch layout helpers (utils/util.py:7-105, data/conformer.py:182-219)
# --------------------------------------------------------------------------
def pad_1d_tokens(values, pad_idx):
    """utils/util.py:7-38 (right pad to the batch max)."""
    size = max(int(v.shape[0]) for v in values)
    res = values[0].new_full((len(values), size), pad_idx)
    for i, v in enumerate(values):
        res[i, : len(v)] = v
    return res


def pad_2d(values, pad_idx):
    """utils/util.py:41-72."""
    size = max(int(v.shape[0]) for v in values)
    res = values[0].new_full((len(values), size, size), pad_idx)
    for i, v in enumerate(values):
        n = len(v)
        res[i, :n, :n] = v
    return res


def pad_coords(values, pad_idx):
    """utils/util.py:75-105."""
    size = max(int(v.shape[0]) for v in values)
    res = values[0].new_full((len(values), size, 3), pad_idx)
    for i, v in enumerate(values):
        res[i, : len(v), :] = v
    return res


def coords2unimol(tokens: np.ndarray, coords: np.ndarray, vocab: int, bos: int = 1, eos: int = 2):
    """data/conformer.py:204-212 given already-indexed atom tokens: BOS/EOS
    wrap, centred coordinates with BOS/EOS at the origin, Euclidean distance
    matrix, edge_type = tok_i * V + tok_j."""
    src_tokens = np.concatenate([[bos], tokens, [eos]]).astype(np.int64)
    c = coords.astype(np.float32)
    c = c - c.mean(axis=0)
    c = np.concatenate([np.zeros((1, 3)), c, np.zeros((1, 3))], axis=0)
    d = np.sqrt(((c[:, None, :] - c[None, :, :]) ** 2).sum(-1))
    et = src_tokens.reshape(-1, 1) * vocab + src_tokens.reshape(1, -1)
    return {
        "src_tokens": src_tokens,
        "src_distance": d.astype(np.float32),
        "src_coord": c.astype(np.float32),
        "src_edge_type": et.astype(np.int64),
    }


# --------------------------------------------------------------------------
# a2-a4: Gaussian basis + projection -> pair bias   (parity unpinned: file
# models/mm_model.py cannot import without Uni-Core; restated from :211-269)
# --------------------------------------------------------------------------
GBF_PI = 3.14159                       # truncated pi, mm_model.py:222
GBF_A = (2 * GBF_PI) ** 0.5


def gaussian(x: Tensor, mean: Tensor, std: Tensor) -> Tensor:
    """mm_model.py:211-224."""
    return torch.exp(-0.5 * (((x - mean) / std) ** 2)) / (GBF_A * std)


def gaussian_layer(dist: Tensor, edge_type: Tensor, P: Params, prefix: str = "gbf.") -> Tensor:
    """GaussianLayer.forward, mm_model.py:254-269.  [B,N,N] -> [B,N,N,K]."""
    mul = P[prefix + "mul.weight"][edge_type]            # [B,N,N,1]
    bias = P[prefix + "bias.weight"][edge_type]
    x = mul * dist.unsqueeze(-1) + bias
    K = P[prefix + "means.weight"].shape[-1]
    x = x.expand(-1, -1, -1, K)
    mean = P[prefix + "means.weight"].float().view(-1)
    std = P[prefix + "stds.weight"].float().view(-1).abs() + 1e-5
    return gaussian(x.float(), mean, std)


def pair_bias(dist: Tensor, edge_type: Tensor, P: Params, bf16: bool = False) -> Tensor:
    """mm_model.py:553-556: gbf -> gbf_proj (Linear-gelu-Linear) -> permute to
    [B*H, N, N]."""
    g = gaussian_layer(dist, edge_type, P)
    # (the pair-bias block keeps bf16 operands in every mode: its fused kernels build basis and hidden in registers)
    h = gelu(linear(g, P["gbf_proj.linear1.weight"], P["gbf_proj.linear1.bias"], bf16, f16ok=False))
    o = linear(h, P["gbf_proj.linear2.weight"], P["gbf_proj.linear2.bias"], bf16, f16ok=False)
    o = o.permute(0, 3, 1, 2).contiguous()
    return o.view(-1, o.size(-2), o.size(-1))


# --------------------------------------------------------------------------
# a5-a6: Uni-Mol pair-bias encoder (Uni-Core semantics; unpinned)
# --------------------------------------------------------------------------
def unimol_layer(x: Tensor, bias: Tensor, P: Params, pre: str, cfg: UniMolCfg,
                 training: bool = False, bf16: bool = False) -> Tuple[Tensor, Tensor, Tensor]:
    """unicore TransformerEncoderLayer (pre-LN) called at models/transformers.py:137-139
    with return_attn=True.  Returns (x, S, P): S = scaled q.k^T + bias, pre-softmax."""
    B, N, D = x.shape
    H = cfg.heads
    hd = D // H
    r = x
    h = layer_norm(x, P[pre + "self_attn_layer_norm.weight"], P[pre + "self_attn_layer_norm.bias"], cfg.ln_eps)
    qkv = linear(h, P[pre + "self_attn.in_proj.weight"], P[pre + "self_attn.in_proj.bias"], bf16)
    qkv = _r(qkv, bf16, "qkv")                # HIP path stores q,k,v as 16-bit values (fp16 by default, bf16 in the bf16 mode)
    q, k, v = qkv.chunk(3, dim=-1)
    scaling = hd ** -0.5

    def heads(t):
        return t.view(B, N, H, hd).transpose(1, 2).contiguous().view(B * H, N, hd)

    q = heads(q) * scaling
    k = heads(k)
    v = heads(v)
    S = torch.bmm(q, k.transpose(1, 2)) + bias
    if bf16 and "s16" in BF16_SITES:
        S = _RoundF16.apply(S)      # pair logits carried between layers as fp16 (what the reference's AMP path does)
    Pm = dropout(torch.softmax(S, dim=-1), cfg.attn_dropout, training)
    o = torch.bmm(Pm, v).view(B, H, N, hd).transpose(1, 2).contiguous().view(B, N, D)
    o = linear(o, P[pre + "self_attn.out_proj.weight"], P[pre + "self_attn.out_proj.bias"], bf16)
    x = r + dropout(o, cfg.dropout, training)
    r = x
    h = layer_norm(x, P[pre + "final_layer_norm.weight"], P[pre + "final_layer_norm.bias"], cfg.ln_eps)
    h = gelu(linear(h, P[pre + "fc1.weight"], P[pre + "fc1.bias"], bf16))
    h = dropout(h, cfg.act_dropout, training)
    h = linear(h, P[pre + "fc2.weight"], P[pre + "fc2.bias"], bf16)
    x = r + dropout(h, cfg.dropout, training)
    return x, S, Pm


def unimol_encoder(emb: Tensor, attn_mask: Tensor, padding_mask: Optional[Tensor], P: Params,
                   cfg: UniMolCfg, pre: str = "encoder.", training: bool = False, bf16: bool = False,
                   with_aux: bool = True):
    """TransformerEncoderWithPair.forward, models/transformers.py:96-183.
    ``attn_mask`` is [B*H,N,N]; NOT modified in place here (the reference fills
    -inf into the caller's tensor at :126)."""
    B, N, _ = emb.shape
    H = cfg.heads
    x = layer_norm(emb, P[pre + "emb_layer_norm.weight"], P[pre + "emb_layer_norm.bias"], cfg.ln_eps)
    x = dropout(x, cfg.emb_dropout, training)
    if padding_mask is not None:
        x = x * (1 - padding_mask.unsqueeze(-1).type_as(x))
    input_attn_mask = attn_mask
    bias = attn_mask
    if bf16 and "s16" in BF16_SITES:
        bias = _RoundF16.apply(bias)
    if padding_mask is not None:
        bias = bias.view(B, H, N, N).masked_fill(padding_mask.view(B, 1, 1, N).bool(), float("-inf")).view(B * H, N, N)
        input_attn_mask = bias          # reference aliasing: the in-place fill is visible in input_attn_mask too
    for i in range(cfg.layers):
        x, bias, _ = unimol_layer(x, bias, P, f"{pre}layers.{i}.", cfg, training, bf16)
    x_pre_final = x
    if (pre + "final_layer_norm.weight") in P:
        x = layer_norm(x, P[pre + "final_layer_norm.weight"], P[pre + "final_layer_norm.bias"], cfg.ln_eps)
    if not with_aux:
        return x, bias
    # aux outputs (:141-181) -- discarded by MM_Model (mm_model.py:559)
    def norm_loss(t, eps=1e-10, tolerance=1.0):
        t = t.float()
        max_norm = t.shape[-1] ** 0.5
        norm = torch.sqrt(torch.sum(t ** 2, dim=-1) + eps)
        return F.relu((norm - max_norm).abs() - tolerance)

    def masked_mean(mask, value, dim=-1, eps=1e-10):
        return (torch.sum(mask * value, dim=dim) / (eps + torch.sum(mask, dim=dim))).mean()

    x_norm = norm_loss(x_pre_final)
    token_mask = 1.0 - padding_mask.float() if padding_mask is not None else torch.ones_like(x_norm)
    x_norm = masked_mean(token_mask, x_norm)
    delta = bias - input_attn_mask                       # -inf - -inf = nan at padded keys ...
    if padding_mask is not None:                          # ... then refilled with 0 (:164)
        delta = delta.view(B, H, N, N).masked_fill(padding_mask.view(B, 1, 1, N).bool(), 0.0).view(B * H, N, N)
    attn = bias.view(B, H, N, N).permute(0, 2, 3, 1).contiguous()
    delta = delta.view(B, H, N, N).permute(0, 2, 3, 1).contiguous()
    pair_mask = token_mask[..., None] * token_mask[..., None, :]
    delta_norm = masked_mean(pair_mask, norm_loss(delta), dim=(-1, -2))
    if (pre + "final_head_layer_norm.weight") in P:
        delta = layer_norm(delta, P[pre + "final_head_layer_norm.weight"], P[pre + "final_head_layer_norm.bias"], cfg.ln_eps)
    return x, attn, delta, x_norm, delta_norm


# --------------------------------------------------------------------------
# a7: RoBERTa tower (HF modeling_roberta.py :75-122 embeddings, :142-155
# position ids, :158-183 attention; post-LN BERT layer)         (pinned: G6)
# --------------------------------------------------------------------------
def roberta_position_ids(input_ids: Tensor, pad_idx: int) -> Tensor:
    """create_position_ids_from_input_ids: cumsum(mask)*mask + pad_idx (int64, bit-exact)."""
    mask = input_ids.ne(pad_idx).int()
    return (torch.cumsum(mask, dim=1).type_as(mask) * mask).long() + pad_idx


def mha(q_in: Tensor, kv_in: Tensor, key_add_mask: Tensor, P: Params, pre: str, heads: int,
        attn_p: float, training: bool, bf16: bool) -> Tensor:
    """BERT-style multi-head attention with separate query/key/value Linears and
    an additive key mask [B,Lk].  Returns the context [B,Lq,D]."""
    B, Lq, D = q_in.shape
    Lk = kv_in.shape[1]
    hd = D // heads
    q = _r(linear(q_in, P[pre + "query.weight"], P[pre + "query.bias"], bf16), bf16, "qkv2")
    k = _r(linear(kv_in, P[pre + "key.weight"], P[pre + "key.bias"], bf16), bf16, "qkv2")
    v = _r(linear(kv_in, P[pre + "value.weight"], P[pre + "value.bias"], bf16), bf16, "qkv2")
    q = q.view(B, Lq, heads, hd).transpose(1, 2)
    k = k.view(B, Lk, heads, hd).transpose(1, 2)
    v = v.view(B, Lk, heads, hd).transpose(1, 2)
    s = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(hd)
    s = s + key_add_mask.view(B, 1, 1, Lk)
    p = dropout(torch.softmax(s, dim=-1), attn_p, training)
    p = _r(p, bf16, "p")
    ctx = torch.matmul(p, v).transpose(1, 2).contiguous().view(B, Lq, D)
    return ctx


def roberta_embeddings(input_ids: Tensor, P: Params, cfg: RobertaCfg, pre: str = "bert.", training=False) -> Tensor:
    e = pre + "embeddings."
    pos = roberta_position_ids(input_ids, cfg.pad_idx)
    # nn.Embedding(padding_idx=...) rows receive no gradient (HF modeling_roberta.py:61,71-73)
    x = F.embedding(input_ids, P[e + "word_embeddings.weight"], padding_idx=cfg.pad_idx) \
        + F.embedding(torch.zeros_like(input_ids), P[e + "token_type_embeddings.weight"])
    x = x + F.embedding(pos, P[e + "position_embeddings.weight"], padding_idx=cfg.pad_idx)
    x = layer_norm(x, P[e + "LayerNorm.weight"], P[e + "LayerNorm.bias"], cfg.ln_eps)
    return dropout(x, cfg.hidden_dropout, training)


def roberta_layer(x: Tensor, key_add_mask: Tensor, P: Params, pre: str, cfg: RobertaCfg,
                  training=False, bf16=False) -> Tensor:
    ctx = mha(x, x, key_add_mask, P, pre + "attention.self.", cfg.heads, cfg.attn_dropout, training, bf16)
    o = linear(ctx, P[pre + "attention.output.dense.weight"], P[pre + "attention.output.dense.bias"], bf16)
    x1 = layer_norm(dropout(o, cfg.hidden_dropout, training) + x,
                    P[pre + "attention.output.LayerNorm.weight"], P[pre + "attention.output.LayerNorm.bias"], cfg.ln_eps)
    i = gelu(linear(x1, P[pre + "intermediate.dense.weight"], P[pre + "intermediate.dense.bias"], bf16))
    o2 = linear(i, P[pre + "output.dense.weight"], P[pre + "output.dense.bias"], bf16)
    return layer_norm(dropout(o2, cfg.hidden_dropout, training) + x1,
                      P[pre + "output.LayerNorm.weight"], P[pre + "output.LayerNorm.bias"], cfg.ln_eps)


def roberta_encoder(input_ids: Tensor, attention_mask: Tensor, P: Params, cfg: RobertaCfg,
                    pre: str = "bert.", training=False, bf16=False) -> Tensor:
    """self.bert(input_ids, attention_mask)[0]  (mm_model.py:562)."""
    x = roberta_embeddings(input_ids, P, cfg, pre, training)
    # masked keys get the dtype minimum added (== P exactly 0 in fp32)
    add = (1.0 - attention_mask.to(x.dtype)) * torch.finfo(torch.float32).min
    for i in range(cfg.layers):
        x = roberta_layer(x, add, P, f"{pre}encoder.layer.{i}.", cfg, training, bf16)
    return x


# --------------------------------------------------------------------------
# a8-a9: InfoNCE head (models/infonce.py)                    (pinned: G1, G2)
# --------------------------------------------------------------------------
def info_nce(query: Tensor, positive_key: Tensor, negative_keys=None, temperature=0.1,
             reduction="mean", negative_mode="unpaired") -> Tensor:
    """models/infonce.py:42-98 (validation :45-67, normalise :70, logits :93, CE :98)."""
    if query.dim() != 2:
        raise ValueError("<query> must have 2 dimensions.")
    if positive_key.dim() != 2:
        raise ValueError("<positive_key> must have 2 dimensions.")
    if negative_keys is not None:
        if negative_mode == "unpaired" and negative_keys.dim() != 2:
            raise ValueError("<negative_keys> must have 2 dimensions if <negative_mode> == 'unpaired'.")
        if negative_mode == "paired" and negative_keys.dim() != 3:
            raise ValueError("<negative_keys> must have 3 dimensions if <negative_mode> == 'paired'.")
    if len(query) != len(positive_key):
        raise ValueError("<query> and <positive_key> must must have the same number of samples.")
    if negative_keys is not None:
        if negative_mode == "paired" and len(query) != len(negative_keys):
            raise ValueError("If negative_mode == 'paired', then <negative_keys> must have the same number of samples as <query>.")
    if query.shape[-1] != positive_key.shape[-1]:
        raise ValueError("Vectors of <query> and <positive_key> should have the same number of components.")
    if negative_keys is not None:
        if query.shape[-1] != negative_keys.shape[-1]:
            raise ValueError("Vectors of <query> and <negative_keys> should have the same number of components.")
    q = F.normalize(query, dim=-1)
    k = F.normalize(positive_key, dim=-1)
    if negative_keys is not None:
        nk = F.normalize(negative_keys, dim=-1)
        pos = torch.sum(q * k, dim=1, keepdim=True)
        if negative_mode == "unpaired":
            neg = q @ nk.transpose(-2, -1)
        else:
            neg = (q.unsqueeze(1) @ nk.transpose(-2, -1)).squeeze(1)
        logits = torch.cat([pos, neg], dim=1)
        labels = torch.zeros(len(logits), dtype=torch.long)
    else:
        logits = q @ k.transpose(-2, -1)
        labels = torch.arange(len(q))
    return (F.cross_entropy(logits / temperature, labels, reduction=reduction)
            + F.cross_entropy(logits.T / temperature, labels, reduction=reduction)) / 2


def infonce_embed(query: Tensor, positive: Tensor, P: Params, pre: str = "infonce.",
                  p: float = 0.1, training=False, bf16=False) -> Tuple[Tensor, Tensor]:
    """InfoNCE.forward up to the two pooled projections (infonce.py:23-33):
    dropout(query) -> per-token Linear-GELU-Linear -> UNMASKED mean over dim 1."""
    xq = dropout(query, p, training)

    def proj(x, name):
        h = F.gelu(linear(x, P[pre + name + ".0.weight"], P[pre + name + ".0.bias"], bf16))
        return _r(linear(h, P[pre + name + ".2.weight"], P[pre + name + ".2.bias"], bf16), bf16, "proj")   # per-token projections stored bf16 (project-then-pool order)

    return proj(xq, "info_proj_query").mean(dim=1), proj(positive, "info_proj_positive").mean(dim=1)


def infonce_forward(query, positive, P, pre="infonce.", temperature=0.1, p=0.1, training=False, bf16=False):
    a, b = infonce_embed(query, positive, P, pre, p, training, bf16)
    return info_nce(a, b, temperature=temperature)


# --------------------------------------------------------------------------
# a10: cross-modal fusion (mm_model.py:379-406, mm_module.py:470-587)  (pinned: G5)
# --------------------------------------------------------------------------
def cross_layer(s1: Tensor, s2: Tensor, s2_add_mask: Tensor, P: Params, pre: str, cfg: CrossCfg,
                training=False, bf16=False) -> Tensor:
    """BertCrossAttentionLayer (mm_module.py:615-626)."""
    ctx = mha(s1, s2, s2_add_mask, P, pre + "attention.self.", cfg.heads, cfg.attn_dropout, training, bf16)
    o = linear(ctx, P[pre + "attention.output.dense.weight"], P[pre + "attention.output.dense.bias"], bf16)
    a = layer_norm(dropout(o, cfg.hidden_dropout, training) + s1,
                   P[pre + "attention.output.LayerNorm.weight"], P[pre + "attention.output.LayerNorm.bias"], cfg.ln_eps)
    i = gelu(linear(a, P[pre + "intermediate.dense.weight"], P[pre + "intermediate.dense.bias"], bf16))
    o2 = linear(i, P[pre + "output.dense.weight"], P[pre + "output.dense.bias"], bf16)
    return layer_norm(dropout(o2, cfg.hidden_dropout, training) + a,
                      P[pre + "output.LayerNorm.weight"], P[pre + "output.LayerNorm.bias"], cfg.ln_eps)


def cross_modal(text_emb: Tensor, graph_emb: Tensor, text_mask: Tensor, graph_mask: Tensor, P: Params,
                cfg: CrossCfg, pre: str = "cross_modal_module.", training=False, bf16=False):
    """CrossAttentionModel.forward (mm_model.py:386-406).  NOTE the caller passes
    (encoder_rep, out_bert, img_mask, attention_mask) so "text" == Uni-Mol atoms."""
    t = dropout(text_emb, cfg.hidden_dropout, training)
    g = dropout(graph_emb, cfg.hidden_dropout, training)
    ext_t = (1.0 - text_mask.float()) * -10000.0
    g2t = cross_layer(g, t, ext_t, P, pre + "graph_attention.layer.0.", cfg, training, bf16)
    ext_g = (1.0 - graph_mask.float()) * -10000.0
    t2g = cross_layer(t, g, ext_g, P, pre + "text_attention.layer.0.", cfg, training, bf16)
    return t2g, g2t


# --------------------------------------------------------------------------
# a12-a13: FDS (models/fds.py, utils/util.py:159-169)              (pinned: G4)
# --------------------------------------------------------------------------
def calibrate_mean_var(matrix, m1, v1, m2, v2, clip_min=0.1, clip_max=10):
    """utils/util.py:159-169 (returns a new tensor except in the partial-column
    branch, which writes the caller's matrix in place exactly like the reference)."""
    if torch.sum(v1) < 1e-10:
        return matrix
    if (v1 == 0.0).any():
        valid = v1 != 0.0
        factor = torch.clamp(v2[valid] / v1[valid], clip_min, clip_max)
        matrix[:, valid] = (matrix[:, valid] - m1[valid]) * torch.sqrt(factor) + m2[valid]
        return matrix
    factor = torch.clamp(v2 / v1, clip_min, clip_max)
    return (matrix - m1) * torch.sqrt(factor) + m2


def fds_kernel_window(kernel: str, ks: int, sigma: float) -> Tensor:
    """FDS._get_kernel_window (fds.py:69-84)."""
    from scipy.ndimage import gaussian_filter1d
    from scipy.signal.windows import triang
    half_ks = (ks - 1) // 2
    if kernel == "gaussian":
        base = np.array([0.0] * half_ks + [1.0] + [0.0] * half_ks, dtype=np.float32)
        w = gaussian_filter1d(base, sigma=sigma) / sum(gaussian_filter1d(base, sigma=sigma))
    elif kernel == "triang":
        w = triang(ks) / sum(triang(ks))
    else:
        lap = lambda x: np.exp(-abs(x) / sigma) / (2.0 * sigma)
        w = np.array(list(map(lap, np.arange(-half_ks, half_ks + 1)))) / sum(map(lap, np.arange(-half_ks, half_ks + 1)))
    return torch.tensor(np.asarray(w), dtype=torch.float32)


def fds_label_bins(labels: Tensor, min_value: float, bin_width: float) -> Tensor:
    """fds.py:125,164: int((value - min_value)//bin_width) per sample, evaluated in
    fp32 tensor arithmetic (python-style floor division).  Returns int64."""
    l0 = labels[:, 0] if labels.dim() > 1 else labels
    return torch.floor_divide(l0.float() - float(min_value), float(bin_width)).to(torch.int64)


class FDSOracle:
    """Functional restatement of models/fds.py:FDS holding the 8 buffers."""

    def __init__(self, feature_dim, min_value, bin_width, bucket_num=100, bucket_start=0, start_update=0,
                 start_smooth=1, kernel="gaussian", ks=5, sigma=2, momentum=0.9):
        self.feature_dim, self.bucket_num, self.bucket_start = feature_dim, bucket_num, bucket_start
        self.start_update, self.start_smooth, self.momentum = start_update, start_smooth, momentum
        self.min_value, self.bin_width = float(min_value), float(bin_width)
        self.kernel_window = fds_kernel_window(kernel, ks, sigma)
        self.half_ks = (ks - 1) // 2
        nb = bucket_num - bucket_start
        self.epoch = torch.zeros(1).fill_(start_update)
        self.running_mean = torch.zeros(nb, feature_dim)
        self.running_var = torch.ones(nb, feature_dim)
        self.running_mean_last_epoch = torch.zeros(nb, feature_dim)
        self.running_var_last_epoch = torch.ones(nb, feature_dim)
        self.smoothed_mean_last_epoch = torch.zeros(nb, feature_dim)
        self.smoothed_var_last_epoch = torch.ones(nb, feature_dim)
        self.num_samples_tracked = torch.zeros(nb)

    @staticmethod
    def bins_from_raw(raw: np.ndarray, bucket_num: int, using_scale: bool):
        """fds.py:47-57: (min_value, bin_width) from the training-CSV column."""
        v = np.array(raw, dtype=np.float64).copy()
        if using_scale:
            v = (v - v.mean()) / v.std()                       # StandardScaler (population std)
            m, s = v.mean(), v.std()                           # anomaly_clean_regression on ndarray: np.std (ddof=0)
            v = v[(v > m - 3 * s) & (v < m + 3 * s)]
        rng = np.max(v) - np.min(v)
        return float(np.min(v)), float(rng / bucket_num)

    def state(self):
        return {k: getattr(self, k).clone() for k in (
            "epoch", "running_mean", "running_var", "running_mean_last_epoch", "running_var_last_epoch",
            "smoothed_mean_last_epoch", "smoothed_var_last_epoch", "num_samples_tracked")}

    def _smooth_stat(self, t: Tensor) -> Tensor:
        x = F.pad(t.unsqueeze(1).permute(2, 1, 0), pad=(self.half_ks, self.half_ks), mode="reflect")
        return F.conv1d(x, self.kernel_window.view(1, 1, -1), padding=0).permute(2, 1, 0).squeeze(1)

    def update_last_epoch_stats(self, epoch):
        """fds.py:86-99,110-114 (note: the reference ALIASES running_* into *_last_epoch)."""
        if epoch == self.epoch + 1:
            self.epoch += 1
            self.running_mean_last_epoch = self.running_mean
            self.running_var_last_epoch = self.running_var
            self.smoothed_mean_last_epoch = self._smooth_stat(self.running_mean_last_epoch)
            self.smoothed_var_last_epoch = self._smooth_stat(self.running_var_last_epoch)

    def _bucket_rows(self, label_bin: Tensor, label: int) -> Tensor:
        if label == self.bucket_start:
            return label_bin <= label
        if label == self.bucket_num - 1:
            return label_bin >= label
        return label_bin == label

    def update_running_stats(self, features: Tensor, labels: Tensor, epoch):
        """fds.py:116-155."""
        if epoch < self.epoch:
            return
        label_bin = fds_label_bins(labels, self.min_value, self.bin_width)
        for label in torch.unique(label_bin).tolist():
            if label > self.bucket_num - 1 or label < self.bucket_start:
                continue
            cur = features[self._bucket_rows(label_bin, label)]
            n = cur.size(0)
            mean = torch.mean(cur, 0)
            var = torch.var(cur, 0, unbiased=(n != 1))
            idx = int(label - self.bucket_start)
            self.num_samples_tracked[idx] += n
            factor = self.momentum if self.momentum is not None else (1 - n / float(self.num_samples_tracked[idx]))
            factor = 0 if epoch == self.start_update else factor
            self.running_mean[idx] = (1 - factor) * mean + factor * self.running_mean[idx]
            self.running_var[idx] = (1 - factor) * var + factor * self.running_var[idx]

    def smooth(self, features: Tensor, labels: Tensor, epoch) -> Tensor:
        """fds.py:157-190.  Out-of-place here (autograd friendly); values equal the
        reference's in-place result."""
        if epoch < self.start_smooth:
            return features
        label_bin = fds_label_bins(labels, self.min_value, self.bin_width)
        out = features
        for label in torch.unique(label_bin).tolist():
            if label > self.bucket_num - 1 or label < self.bucket_start:
                continue
            rows = self._bucket_rows(label_bin, label)
            idx = int(label - self.bucket_start)
            new = calibrate_mean_var(out[rows].clone(), self.running_mean_last_epoch[idx], self.running_var_last_epoch[idx],
                                     self.smoothed_mean_last_epoch[idx], self.smoothed_var_last_epoch[idx])
            out = out.clone()
            out[rows] = new
        return out


# --------------------------------------------------------------------------
# a15-a17: ConR / SupCon (models/contrastive.py)                  (pinned: G3)
# --------------------------------------------------------------------------
def _ct_core(feature, pos_i, neg_i, pushing_w, denom, t):
    q = F.normalize(feature.reshape(feature.shape[0], -1), dim=1)
    prod = (q @ q.T) / t
    pos = prod * pos_i
    neg = prod * neg_i
    neg_exp_dot = (pushing_w * torch.exp(neg) * neg_i).sum(1)
    no_neg_flag = neg_i.sum(1).bool()
    loss = ((-torch.log(torch.exp(pos) / (torch.exp(pos).sum(1) + neg_exp_dot).unsqueeze(-1)) * pos_i).sum(1) / denom)
    return (loss * no_neg_flag).unsqueeze(-1).mean()


def ct_regress(feature, depth, output, weights=None, w=0.2, t=0.07, e=0.01):
    """CT_Regress, contrastive.py:3-59."""
    depth = depth.reshape(depth.shape[0], -1)
    l = torch.mean(depth, dim=1).unsqueeze(-1)
    output = output.reshape(output.shape[0], -1)
    p = torch.mean(output, dim=1).unsqueeze(-1)
    l_dist = torch.abs(l - l.T)
    p_dist = torch.abs(p - p.T)
    le = l_dist.le(w)
    pos_i = le.clone()
    neg_i = (~le) * p_dist.le(w)
    pos_i.fill_diagonal_(False)
    if weights is None:
        weights = torch.ones_like(l_dist)
    weights = torch.mean(weights.reshape(weights.shape[0], -1), dim=1).unsqueeze(-1)
    pushing_w = l_dist * weights * e
    denom = le.sum(1)
    return _ct_core(feature, pos_i, neg_i, pushing_w, denom, t)


def ct_single(feature, depth, output=None, weights=None, w=0.2, t=0.07, e=0.2, lamda=1):
    """CT_Single, contrastive.py:62-112 (weights default tensor([1]); a [B] vector
    broadcasts along keys j)."""
    depth = depth.reshape(depth.shape[0], 1)
    l_dist = torch.abs(depth - depth.T)
    pos_i = l_dist.eq(0)
    neg_i = ~l_dist.eq(0)
    pos_i = pos_i.clone()
    pos_i.fill_diagonal_(False)
    if weights is None:
        weights = torch.tensor([1])
    denom = pos_i.sum(1)
    denom = torch.where(denom == 0, torch.ones_like(denom), denom)
    return _ct_core(feature, pos_i, neg_i, weights, denom, t)


def ct_multi(feature, depth, output=None, weights=None, w=0.2, t=0.07, e=0.2, coef=1):
    """CT_Multi, contrastive.py:114-169 (the B^2 Python loop vectorised)."""
    depth = depth.reshape(depth.shape[0], -1)
    C = depth.shape[1]
    sim = (depth.unsqueeze(1) == depth.unsqueeze(0)).sum(-1).to(torch.float32) / C
    thr = coef / C
    pos_i = sim.ge(thr).clone()
    neg_i = ~sim.ge(thr)
    pos_i.fill_diagonal_(False)
    pw = weights if weights is not None else torch.ones(())
    denom = pos_i.sum(1)
    denom = torch.where(denom == 0, torch.ones_like(denom), denom)
    return _ct_core(feature, pos_i, neg_i, pw, denom, t)


# --------------------------------------------------------------------------
# a1, a11, a14 + MM_Model.forward (mm_model.py:526-618)
# --------------------------------------------------------------------------
def classification_head(x, P, pre="classification_head.", p=0.2, training=False, bf16=False):
    """mm_model.py:44-84."""
    x = dropout(x, p, training)
    x = torch.tanh(linear(x, P[pre + "dense.weight"], P[pre + "dense.bias"], bf16))
    x = dropout(x, p, training)
    return linear(x, P[pre + "out_proj.weight"], P[pre + "out_proj.bias"], bf16)


def mm_features(batch: Dict[str, Tensor], P: Params, cfg: ModelCfg, training=False, bf16=False):
    """mm_model.py:545-576 -> (encoder_rep, out_bert, pooled)."""
    u = cfg.unimol
    src_tokens = batch["src_tokens"]
    padding_mask = src_tokens.eq(u.pad_idx)
    img_mask = ~padding_mask
    attention_mask = batch["attention_mask"].bool()
    pm = padding_mask if padding_mask.any() else None
    x = F.embedding(src_tokens, P["embed_tokens.weight"], padding_idx=u.pad_idx)   # mm_model.py:439-441
    bias = pair_bias(batch["src_distance"], batch["src_edge_type"], P, bf16)
    enc, _ = unimol_encoder(x, bias, pm, P, u, "encoder.", training, bf16, with_aux=False)
    bert = roberta_encoder(batch["input_ids"], batch["attention_mask"], P, cfg.roberta, "bert.", training, bf16)
    t2g, g2t = cross_modal(enc, bert, img_mask, attention_mask, P, cfg.cross, "cross_modal_module.", training, bf16)
    t2g = t2g * img_mask.unsqueeze(-1)
    g2t = g2t * attention_mask.unsqueeze(-1)
    final = torch.cat((t2g, g2t), dim=1)
    pooled = final.sum(dim=1) / (img_mask.sum(dim=1).view(-1, 1) + attention_mask.sum(dim=1).view(-1, 1))
    return enc, bert, pooled


def mm_forward(batch, P, cfg: ModelCfg, net_target=None, weights=None, use_weight=False, fds: Optional[FDSOracle] = None,
               epoch=0, training=False, bf16=False):
    """Full MM_Model.forward with return_infonce_loss=True, return_ct_loss=True.
    Returns dict(logits, infonce, ct, pooled, enc, bert)."""
    enc, bert, pooled = mm_features(batch, P, cfg, training, bf16)
    infonce = infonce_forward(enc, bert, P, "infonce.", cfg.infonce_temp, cfg.infonce_dropout, training, bf16)
    feats = pooled
    if training and fds is not None and epoch >= fds.start_smooth and cfg.task == "regression":
        feats = fds.smooth(feats, net_target, epoch)            # reference aliasing: CT sees the smoothed tensor
    logits = classification_head(feats, P, "classification_head.", cfg.unimol.pooler_dropout, training, False)  # head runs fp32 on the HIP path
    ct = None
    if net_target is not None:
        kw = dict(w=cfg.ct_w)
        if use_weight:
            kw["weights"] = weights
        if cfg.task == "classification":
            ct = ct_single(feats, net_target, logits, **kw)
        elif cfg.task == "multilabel_classification":
            ct = ct_multi(feats, net_target, logits, **kw)
        else:
            ct = ct_regress(feats, net_target, logits, **kw)
    return dict(logits=logits, infonce=infonce, ct=ct, pooled=feats, enc=enc, bert=bert)


def task_loss(logits, target, task):
    """models/nnmodel.py:24-34, models/loss.py:278-289."""
    if task == "regression":
        return F.mse_loss(logits, target.float())
    if task == "multilabel_classification":          # LOSS_RREGISTER['multilabel_classification']['bce'] = nn.BCEWithLogitsLoss()
        return F.binary_cross_entropy_with_logits(logits, target.float())
    return F.cross_entropy(logits, target.flatten().long())


def step_loss(out, target, task, alpha=1.0, beta=0.1):
    """tasks/trainer.py:192-193: alpha*task + beta*infonce + beta*ct."""
    tl = task_loss(out["logits"], target, task)
    return alpha * tl + beta * out["infonce"] + beta * out["ct"], tl


# --------------------------------------------------------------------------
# random-init parameters of the reference architecture (init_bert_params-like)
# --------------------------------------------------------------------------
def init_params(cfg: ModelCfg, seed: int = 0, std: float = 0.02) -> Params:
    g = torch.Generator().manual_seed(seed)
    P: Params = {}

    def lin(name, out_f, in_f):
        P[name + ".weight"] = torch.randn(out_f, in_f, generator=g) * std
        P[name + ".bias"] = torch.randn(out_f, generator=g) * std

    def ln(name, d):
        P[name + ".weight"] = 1.0 + 0.1 * torch.randn(d, generator=g)
        P[name + ".bias"] = 0.1 * torch.randn(d, generator=g)

    u, r, c = cfg.unimol, cfg.roberta, cfg.cross
    P["embed_tokens.weight"] = torch.randn(u.vocab, u.dim, generator=g) * std
    P["embed_tokens.weight"][u.pad_idx] = 0
    E = u.vocab * u.vocab
    P["gbf.means.weight"] = torch.rand(1, u.K, generator=g) * 3
    P["gbf.stds.weight"] = torch.rand(1, u.K, generator=g) * 3
    P["gbf.mul.weight"] = 1.0 + 0.1 * torch.randn(E, 1, generator=g)
    P["gbf.bias.weight"] = 0.1 * torch.randn(E, 1, generator=g)
    lin("gbf_proj.linear1", u.K, u.K)
    lin("gbf_proj.linear2", u.heads, u.K)
    ln("encoder.emb_layer_norm", u.dim)
    ln("encoder.final_layer_norm", u.dim)
    for i in range(u.layers):
        p = f"encoder.layers.{i}."
        lin(p + "self_attn.in_proj", 3 * u.dim, u.dim)
        lin(p + "self_attn.out_proj", u.dim, u.dim)
        ln(p + "self_attn_layer_norm", u.dim)
        lin(p + "fc1", u.ffn, u.dim)
        lin(p + "fc2", u.dim, u.ffn)
        ln(p + "final_layer_norm", u.dim)
    e = "bert.embeddings."
    P[e + "word_embeddings.weight"] = torch.randn(r.vocab, r.dim, generator=g) * std
    P[e + "word_embeddings.weight"][r.pad_idx] = 0
    P[e + "position_embeddings.weight"] = torch.randn(r.max_pos, r.dim, generator=g) * std
    P[e + "position_embeddings.weight"][r.pad_idx] = 0
    P[e + "token_type_embeddings.weight"] = torch.randn(r.type_vocab, r.dim, generator=g) * std
    ln(e + "LayerNorm", r.dim)

    def bert_layer(p, dim, ffn):
        for n in ("query", "key", "value"):
            lin(p + "attention.self." + n, dim, dim)
        lin(p + "attention.output.dense", dim, dim)
        ln(p + "attention.output.LayerNorm", dim)
        lin(p + "intermediate.dense", ffn, dim)
        lin(p + "output.dense", dim, ffn)
        ln(p + "output.LayerNorm", dim)

    for i in range(r.layers):
        bert_layer(f"bert.encoder.layer.{i}.", r.dim, r.ffn)
    for side in ("text_attention", "graph_attention"):
        bert_layer(f"cross_modal_module.{side}.layer.0.", c.dim, c.ffn)
    for n in ("info_proj_query", "info_proj_positive"):
        lin(f"infonce.{n}.0", c.dim, c.dim)
        lin(f"infonce.{n}.2", cfg.infonce_dim, c.dim)
    lin("classification_head.dense", u.dim, c.dim)
    lin("classification_head.out_proj", cfg.output_dim, u.dim)
    return P


# --------------------------------------------------------------------------
# synthetic batches (SURVEY.md section 8d "C-main")
# --------------------------------------------------------------------------
def synth_batch(B: int, max_atoms: int, max_tokens: int, cfg: ModelCfg, seed: int = 1234, ragged: bool = False,
                n_labels: int = 1) -> Tuple[Dict[str, Tensor], Tensor]:
    """Seeded synthetic collated batch in the reference's layout (a0):
    tokens [BOS]+atoms+[EOS] right-padded with pad_idx; distances Euclidean with
    BOS/EOS at the origin; edge_type = tok_i*V+tok_j padded with pad_idx;
    SMILES ids <s> ... </s> right-padded with the RoBERTa pad id."""
    rng = np.random.default_rng(seed)
    u, r = cfg.unimol, cfg.roberta
    toks, dists, ets, ids = [], [], [], []
    elem_p = np.zeros(u.vocab)
    elem_p[8] = 0.5; elem_p[4] = 0.3; elem_p[5] = 0.075; elem_p[6] = 0.075     # H, C, N, O
    rest = [i for i in range(4, u.vocab - 1) if i not in (4, 5, 6, 8)]
    elem_p[rest] = 0.05 / len(rest)
    for _ in range(B):
        if ragged:
            # clamp(round(N(48,20^2)),8,128) at max_atoms=128, scaled for smaller test shapes
            na = int(np.clip(round(rng.normal(0.375 * max_atoms, 0.16 * max_atoms)), max(2, max_atoms // 16), max_atoms))
            nt = int(np.clip(round(1.6 * na * 0.5), min(8, max(4, max_tokens // 4)), max_tokens))
        else:
            na, nt = max_atoms, max_tokens
        a = rng.choice(u.vocab, size=na, p=elem_p)
        c = rng.normal(0, 3.0, size=(na, 3))
        d = coords2unimol(a, c, u.vocab)
        toks.append(torch.from_numpy(d["src_tokens"]))
        dists.append(torch.from_numpy(d["src_distance"]))
        ets.append(torch.from_numpy(d["src_edge_type"]))
        body = rng.integers(4, r.vocab, size=nt - 2)
        ids.append(torch.from_numpy(np.concatenate([[0], body, [2]]).astype(np.int64)))
    input_ids = pad_1d_tokens(ids, r.pad_idx)
    batch = {
        "src_tokens": pad_1d_tokens(toks, u.pad_idx),
        "src_distance": pad_2d(dists, 0.0),
        "src_edge_type": pad_2d(ets, u.pad_idx),
        "input_ids": input_ids,
        "attention_mask": input_ids.ne(r.pad_idx).long(),
    }
    if cfg.task == "regression":
        label = torch.from_numpy(rng.normal(0, 1, size=(B, 1)).astype(np.float32))
    elif cfg.task == "multilabel_classification":
        label = torch.from_numpy((rng.random((B, n_labels)) < 0.2).astype(np.int64))
    else:
        label = torch.from_numpy((rng.random((B, 1)) < 0.2).astype(np.int64))
    return batch, label
