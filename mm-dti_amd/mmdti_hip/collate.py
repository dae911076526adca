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
