"""Host-side batch layout: ragged per-molecule arrays -> the right-padded tensors the kernels read.

Layout contract (the reference's ``utils/util.py:7-105`` helpers and ``MM_Model.batch_collate_fn``,
models/mm_model.py:645-682; pinned bit-exact by tests/golden/g7_pad.npz and g9_collate.npz):
  * every field is right-padded to the batch maximum, the output keeps the dtype of the first sample;
  * ``src_tokens`` [B,N] and ``src_edge_type`` [B,N,N] pad with the dictionary's pad index (NOT pad*V+pad for edges),
    ``src_distance`` [B,N,N] and ``src_coord`` [B,N,3] with 0.0; ``weights`` are stacked as given;
  * ``smile`` strings go through the HF tokenizer (padding=True, truncation=True) and come back as ``input_ids`` /
    ``attention_mask`` at the END of the dict;
  * labels that cannot be stacked into one tensor give ``None``.
This is pure indexing on the host; nothing here touches the device.
"""
import numpy as np
import torch


def right_pad(values, fill, square=False, tail=()):
    """values: list of tensors whose leading extent (both leading extents when ``square``) is ragged -> one tensor
    [len(values), n_max(, n_max), *tail] filled with ``fill`` outside each sample's extent."""
    n_max = max(int(v.shape[0]) for v in values)
    shape = (len(values), n_max) + ((n_max,) if square else ()) + tuple(tail)
    out = values[0].new_full(shape, fill)
    for row, v in zip(out, values):
        n = int(v.shape[0])
        if square:
            row[:n, :n] = v
        else:
            row[:n] = v
    return out


PAD_IDX = object()          # "fill with the model's dictionary pad index"
FIELD_RULES = {             # key -> (dtype, fill, square, tail)
    'src_tokens': (torch.int64, PAD_IDX, False, ()),
    'src_edge_type': (torch.int64, PAD_IDX, True, ()),
    'src_distance': (torch.float32, 0.0, True, ()),
    'src_coord': (torch.float32, 0.0, False, (3,)),
}


def collate_field(key, column, padding_idx):
    """-> padded tensor for a known key, stacked tensor for 'weights', None for a key without a rule."""
    if key == 'weights':
        return torch.tensor(list(column))
    rule = FIELD_RULES.get(key)
    if rule is None:
        return None
    dtype, fill, square, tail = rule
    return right_pad([torch.tensor(v).to(dtype) for v in column], padding_idx if fill is PAD_IDX else fill, square, tail)


def stack_labels(samples):
    try:
        return torch.tensor(np.asarray([s[1] for s in samples]))
    except Exception:
        return None


def tokenize(tokenizer, smiles, truncation=True):
    enc = tokenizer(list(smiles), padding=True, truncation=truncation, return_tensors="pt")
    return enc['input_ids'], enc['attention_mask']


def collate_batch(samples, padding_idx, tokenizer):
    """The whole of ``MM_Model.batch_collate_fn`` (models/mm_model.py:645-682) as a free function: list of
    (feature dict, label) -> (batch dict, label tensor | None).  A key without a layout rule re-uses the previous field's
    value, as the reference's loop variable does (and raises if it comes first)."""
    feats = [s[0] for s in samples]
    batch, last = {}, None
    for key in feats[0]:
        if key == 'smile':
            continue
        v = collate_field(key, (f[key] for f in feats), padding_idx)
        if v is None:
            if last is None:
                raise UnboundLocalError(f"batch_collate_fn: no layout rule for the first feature key {key!r}")
            v = last
        batch[key] = last = v
    if 'smile' in feats[0]:
        batch['input_ids'], batch['attention_mask'] = tokenize(tokenizer, (f['smile'] for f in feats))
    return batch, stack_labels(samples)


# -------------------------------------------------------------------------------------------------- device payload (8f-3)
NEVER_CONSUMED = ('src_coord',)     # collated by the reference, swallowed by **kwargs in MM_Model.forward (mm_model.py:540)
# stay on the host: MM_Model reads them to pick kernels and layouts without a device sync
HOST_FIELDS = ('atom_counts', 'token_counts', 'token_pad_id', 'packable')
INT16_MAX = 32767


def atom_counts(src_tokens, pad_idx=0):
    """[B] int32: position of each molecule's last non-pad token + 1 (its real length for a right-padded batch; with holes in
    the mask, still an upper bound of every real key's index -- which is all the ragged pair kernels rely on)."""
    real = src_tokens.ne(pad_idx)
    pos = torch.arange(1, src_tokens.shape[1] + 1, dtype=torch.int64, device=src_tokens.device)
    return (real.to(torch.int64) * pos).amax(dim=1).to(torch.int32)


def to_device(net_input, device, non_blocking=True):
    """.to(device) for every field of a collated batch except the host-side ones."""
    return {k: (v if k in HOST_FIELDS else v.to(device, non_blocking=non_blocking)) for k, v in net_input.items()}


def device_payload(net_input, n_edge_types=None, pad_idx=0):
    """What of a collated batch actually has to cross PCIe: drops the fields no kernel reads and narrows
    ``src_edge_type`` from int64 to int16 (8 -> 2 bytes per atom pair; the pair-bias kernels take either width) when every
    index fits -- ``n_edge_types`` (= len(dictionary)**2, 961 for the reference's dictionary) decides without a scan, else
    the tensor's own min / max do.  Values are unchanged; host tensors in, host tensors out."""
    out = {k: v for k, v in net_input.items() if k not in NEVER_CONSUMED}
    et = out.get('src_edge_type')
    if et is not None and et.dtype == torch.int64:
        if n_edge_types is not None:
            fits = n_edge_types <= INT16_MAX + 1
        else:
            fits = et.numel() == 0 or (int(et.min()) >= -INT16_MAX - 1 and int(et.max()) <= INT16_MAX)
        if fits:
            out['src_edge_type'] = et.to(torch.int16)
    st = out.get('src_tokens')
    if st is not None and st.device.type == 'cpu' and 'atom_counts' not in out:
        out['atom_counts'] = atom_counts(st, pad_idx)
    out.update(packing_fields(out, pad_idx))
    return out


def packing_fields(batch, pad_idx=0):
    """Host-side facts that let MM_Model run a ragged batch on PACKED token rows (packing.py): ``token_counts`` ([B] int32 SMILES
    lengths), ``token_pad_id`` (the one id every masked SMILES slot holds, -1 when nothing is masked) and ``packable`` -- True iff
    both sides are right-padded as the reference collates them (mm_model.py:645-682, HF ``padding=True``): atom tokens without a
    pad hole, attention masks that are non-empty prefixes of ones, one id in all masked slots.  {} when it cannot be decided here
    (device tensors, fields missing)."""
    from .packing import right_padded_lengths
    st, ids, am = batch.get('src_tokens'), batch.get('input_ids'), batch.get('attention_mask')
    if any(t is None or not torch.is_tensor(t) or t.device.type != 'cpu' for t in (st, ids, am)) or 'token_counts' in batch:
        return {}
    atoms = right_padded_lengths(st.ne(pad_idx))
    toks = right_padded_lengths(am.ne(0))
    if atoms is None or toks is None:
        return {'packable': False}
    masked = ids[am.eq(0)]
    pad_id = -1
    if masked.numel():
        pad_id = int(masked[0])
        if not bool((masked == pad_id).all()):
            return {'packable': False}
    return {'token_counts': toks.to(torch.int32), 'token_pad_id': pad_id, 'packable': True}


class HostCollate:
    """``collate_fn`` for a ``DataLoader`` with worker processes: the model's collate (same tensors as
    ``model.batch_collate_fn``) followed by :func:`device_payload`, holding only the pad index, the tokenizer and the
    edge-type count -- so the workers never receive a copy of the model."""

    def __init__(self, padding_idx, tokenizer, n_edge_types=None, narrow=True):
        self.padding_idx, self.tokenizer, self.n_edge_types, self.narrow = padding_idx, tokenizer, n_edge_types, narrow

    @classmethod
    def of(cls, model, narrow=True):
        n = None
        d = getattr(model, 'dictionary', None)
        if d is not None:
            n = len(d) * len(d)
        return cls(model.padding_idx, getattr(model, 'tokenizer', None), n, narrow)

    def __call__(self, samples):
        batch, label = collate_batch(samples, self.padding_idx, self.tokenizer)
        return (device_payload(batch, self.n_edge_types, self.padding_idx) if self.narrow else batch), label
