"""Packed (variable-length) token layout of a right-padded batch -- host-side index arithmetic only.

The reference pads every batch to its longest molecule / SMILES string (``models/mm_model.py:645-682``; HF tokenizer
``padding=True``) and then computes on every padded row: ``TransformerEncoderWithPair`` zeroes the padded rows
(``models/transformers.py:114-118``) but runs them through all 15 layers as queries, RoBERTa does the same with its pad
embedding, and ``InfoNCE.forward`` averages over ALL positions (``models/infonce.py:32-33``).  At dropout 0 every padded row
of one molecule is the SAME row in every layer ("Ragged batches"):

  * tower 1: the input row is zero, its bias row is the constant that ``src_distance`` = 0 / ``src_edge_type`` = pad index
    give (``mm_model.py:657-661``), its keys are the molecule's real tokens (padded keys carry -inf);
  * tower 2: pad token embedding + position ``padding_idx`` for every masked slot, keys = the real tokens.

So the kernels compute ONE representative pad row per sequence and the unmasked InfoNCE mean weights it by the number of padded
positions: ``mean = (sum_real x_i + n_pad * x_pad) / S``.  Cross-modal fusion and the masked pool never read padded rows
(masked keys have probability exactly 0; padded query rows are zeroed by ``mm_model.py:572-573`` before the pooled sum).

Row layout of a tower: sequence ``b`` owns rows ``[off[b], off[b+1])`` -- its ``n_real[b]`` real tokens in order, then (iff
``n_real[b] < S``) the representative pad row, which is position ``n_real[b]`` of the padded tensor (the first padded slot:
gathering any per-position input there picks up the pad value).

With dropout ON the reference draws an independent mask for every padded row; one weighted row is then equal in expectation
only -- ``MM_Model(strict_reference=True)`` keeps the padded computation.
"""
from __future__ import annotations

import torch


class PackedRows:
    """counts: [B] integers on the HOST (real length of every sequence, 1 <= counts[b] <= S); S: the padded length."""

    def __init__(self, counts, S: int, device=None):
        c = torch.as_tensor(counts, device="cpu").to(torch.int64).reshape(-1)
        if c.numel() == 0 or int(c.min()) < 1 or int(c.max()) > S:
            raise ValueError(f"PackedRows: lengths must lie in [1, {S}] (got min {int(c.min()) if c.numel() else None}, max {int(c.max()) if c.numel() else None})")
        B = c.numel()
        rows = c + (c < S).to(torch.int64)                     # + the representative pad row
        off = torch.zeros(B + 1, dtype=torch.int64)
        torch.cumsum(rows, 0, out=off[1:])
        M = int(off[-1])
        seq = torch.repeat_interleave(torch.arange(B, dtype=torch.int64), rows)          # [M] sequence of each packed row
        local = torch.arange(M, dtype=torch.int64) - off[:-1][seq]                        # [M] position inside its sequence
        self.B, self.S, self.M = B, int(S), M
        self.max_rows = int(rows.max())
        self.counts_host, self.rows_host, self.off_host = c, rows, off
        self.gather_host = seq * S + local                     # flat index into the padded [B*S] arrays
        self.seq_host, self.local_host = seq, local
        self.n_pad_rows = int((c < S).sum())
        # device images (filled by to(): ONE int32 upload + one int64 upload)
        self.off = self.n_real = self.row_seq = self.gather = None
        if device is not None:
            self.to(device)

    def to(self, device):
        B, M = self.B, self.M
        i32 = torch.cat([self.off_host, self.counts_host, self.seq_host]).to(torch.int32)
        from .ops import upload          # (pinned staging: a pageable non_blocking copy stalls the host behind the stream's queue)
        d32 = upload(i32, device)
        self.off, self.n_real, self.row_seq = d32[:B + 1], d32[B + 1:2 * B + 1], d32[2 * B + 1:]
        self.gather = upload(self.gather_host, device)
        return self

    # ---- glue for API boundaries and tests (plain indexing, not on the hot path)
    def pack(self, x):
        """[B, S, ...] -> [M, ...]: the packed rows of a padded tensor."""
        return x.reshape(self.B * self.S, *x.shape[2:])[self.gather.to(x.device)]

    def unpack(self, xp):
        """[M, ...] -> [B, S, ...]: every padded position receives its sequence's representative pad row."""
        pos = torch.arange(self.S, dtype=torch.int64).view(1, -1)
        idx = self.off_host[:-1].view(-1, 1) + torch.minimum(pos, self.counts_host.view(-1, 1))
        return xp[idx.reshape(-1).to(xp.device)].view(self.B, self.S, *xp.shape[1:])

    def pad_weights(self):
        """[M] fp32 on the host: 1 for a real row, S - n_real for the representative pad row (the InfoNCE mean's weights x S)."""
        w = torch.ones(self.M, dtype=torch.float32)
        is_pad = self.local_host >= self.counts_host[self.seq_host]
        w[is_pad] = (self.S - self.counts_host[self.seq_host][is_pad]).to(torch.float32)
        return w


def right_padded_lengths(mask_or_real):
    """[B, S] 0/1 (1 = real) on the HOST -> [B] int64 lengths if every row is a non-empty prefix of ones, else None."""
    m = mask_or_real.to(torch.int64)
    n = m.sum(dim=1)
    pos = torch.arange(1, m.shape[1] + 1, dtype=torch.int64)
    last = (m * pos).amax(dim=1)
    if bool((n == last).all()) and int(n.min()) >= 1:
        return n
    return None
