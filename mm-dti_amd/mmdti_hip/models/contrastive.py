"""Drop-in for the reference's ``models/contrastive.py``: CT_Regress (ConR), CT_Single (SupCon), CT_Multi.

Same free-function names, argument order and defaults as /root/reference/models/contrastive.py:3,62,114.  The B x B
masks, the exp sums and the gradient are produced by two gfx950 kernels (no Python loop over the batch; the reference
runs ``for i in range(B): pos_i[i][i] = 0`` and, in CT_Multi, a B^2 Python loop on the CPU).

Reference quirks kept: masked-out logits contribute exp(0)=1 to the positive denominator (:35,:52); ConR's ``denom``
counts the anchor itself (:49); SupCon/Multi weights broadcast along KEYS (:95,:150); ``output`` only shapes ConR's
hard-negative mask and gets no gradient; ``lamda`` is accepted and ignored.
"""
import torch

from .. import ops
from ..functional import CTLossFn


def _f32(t):
    return t.detach().to(torch.float32).contiguous()


def _weights(weights, B, per_anchor, device=None):
    if weights is None:
        return None
    if not per_anchor and weights.numel() == 1:
        # (the reference's default argument is the HOST tensor([1]): decide here, on the host -- moving it to the device and
        #  reading it back was a pageable copy plus an .item(), i.e. two full device synchronisations in the middle of every step)
        if weights.device.type == "cpu":
            v = float(weights)
            return None if v == 1.0 else torch.full((B,), v, device=device, dtype=torch.float32)
        return weights.detach().to(torch.float32).reshape(1).expand(B).contiguous()
    if device is not None:
        weights = weights.to(device)
    w = weights.detach().to(torch.float32)
    if per_anchor:
        # CT_Regress: mean over trailing dims -> one weight per ANCHOR row (:39-40)
        return w.reshape(w.shape[0], -1).mean(dim=1).contiguous()
    w = w.reshape(-1)
    if w.numel() != B:
        raise ValueError(f"weights must have 1 or {B} elements, got {w.numel()}")
    return w.contiguous()


def CT_Regress(feature, depth, output, weights=None, w=0.2, t=0.07, e=0.01):
    B = feature.shape[0]
    lab = _f32(depth).reshape(B, -1).mean(dim=1).contiguous()
    pred = _f32(output).reshape(B, -1).mean(dim=1).contiguous()
    wt = None
    if weights is not None:
        wt = _weights(weights, B, per_anchor=True, device=feature.device)
    return CTLossFn.apply(feature.float(), ops.CT_REGRESS, lab, None, pred, wt, float(w), float(t), float(e), 1.0)


def CT_Single(feature, depth, output, weights=torch.tensor([1]), w=0.2, t=0.07, e=0.2, lamda=1):
    B = feature.shape[0]
    lab = _f32(depth).reshape(B, -1)
    if lab.shape[1] != 1:
        raise ValueError("CT_Single expects one label per sample")
    wt = _weights(weights, B, per_anchor=False, device=feature.device)
    return CTLossFn.apply(feature.float(), ops.CT_SINGLE, lab.reshape(B).contiguous(), None, None, wt, float(w), float(t), float(e), 1.0)


def CT_Multi(feature, depth, output, weights=None, w=0.2, t=0.07, e=0.2, coef=1):
    B = feature.shape[0]
    lab = depth.detach().reshape(B, -1).to(torch.int64).contiguous()
    wt = None if weights is None else _weights(weights, B, per_anchor=False, device=feature.device)
    return CTLossFn.apply(feature.float(), ops.CT_MULTI, None, lab, None, wt, float(w), float(t), float(e), float(coef))
