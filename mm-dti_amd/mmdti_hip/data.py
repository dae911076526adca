"""Host-side input pipeline helpers (the step's caller side).

The reference batches molecules in dataset order and right-pads every batch to its longest member
(``MM_Model.batch_collate_fn``, models/mm_model.py:645-682; ``pad_1d_tokens`` / ``pad_2d``, utils/util.py:7-72).  The
step's cost is set by the PADDED shape -- pair tensors grow with N_max^2, attention with L_max^2 -- so batching molecules
of similar size is the largest lever a real dataset offers.  ``LengthBucketBatchSampler`` does that without changing what
a batch is (same collate, same tensors, same semantics for InfoNCE / SupCon: the negatives of a step are its batch).
"""
from __future__ import annotations

from typing import Iterator, List, Sequence

HOST_FIELDS = ('atom_counts', 'token_counts', 'token_pad_id', 'packable')      # (collate.HOST_FIELDS; repeated here so that this module keeps importing without torch)

import numpy as np


def padded_cost(atoms: Sequence[int], tokens: Sequence[int]) -> float:
    """Relative cost of one padded batch: pair work ~ B * N_max^2 * 15 layers, tower 2 ~ B * L_max * (const + L_max)."""
    b = len(atoms)
    n, l = max(atoms) + 2, max(tokens)
    return b * (15.0 * n * n + 6.0 * l * (12.0 * 512 / 64 + l) / 8.0)


class LengthBucketBatchSampler:
    """Batch sampler (yields lists of dataset indices) that groups molecules of similar (atom count, SMILES token count).

    * every index appears exactly once per epoch (``drop_last=False``, single process); with ``drop_last=True`` the
      ``n % batch_size`` molecules left out are drawn AT RANDOM each epoch (as the reference's shuffled ``drop_last``
      DataLoader does, tasks/trainer.py:143-150) -- never systematically the largest ones;
    * ``shuffle``: the ORDER of batches and the membership within a size neighbourhood are reshuffled every epoch from
      ``seed + epoch`` (call ``set_epoch``) -- samples still meet different partners from epoch to epoch, which the
      in-batch contrastive losses need;
    * data parallel: ``rank`` / ``world`` deal whole batches round-robin and trim to a common count, so that every rank
      runs the same number of steps (the gradient all-reduce is collective).  Every dealt batch is FULL: the InfoNCE
      all-gather (parallel.GlobalNegatives) needs the same B_loc on all ranks at a step, so with ``world > 1`` a random
      remainder is left out each epoch even when ``drop_last=False``.
    """

    def __init__(self, atom_counts: Sequence[int], token_counts: Sequence[int], batch_size: int, shuffle: bool = True, seed: int = 0,
                 drop_last: bool = False, neighbourhood: int = 8, rank: int = 0, world: int = 1):
        assert len(atom_counts) == len(token_counts) and batch_size > 0 and neighbourhood >= 1 and 0 <= rank < world
        self.atoms = np.asarray(atom_counts, dtype=np.int64)
        self.tokens = np.asarray(token_counts, dtype=np.int64)
        self.batch_size, self.shuffle, self.seed, self.drop_last = batch_size, shuffle, seed, drop_last
        self.neighbourhood, self.rank, self.world = neighbourhood, rank, world
        self.epoch = 0

    def set_epoch(self, epoch: int):
        self.epoch = epoch

    def _batches(self) -> List[List[int]]:
        rng = np.random.default_rng(self.seed + self.epoch)
        n = len(self.atoms)
        # sort by the dominant cost term (atoms, then tokens); jitter within a neighbourhood of batches so that membership
        # changes between epochs while sizes stay close
        keep = np.arange(n)
        if (self.drop_last or self.world > 1) and n % self.batch_size:
            # leave out a random remainder BEFORE sorting (dropping the tail of the size-sorted order would always drop the
            # largest molecules)
            keep = np.sort(rng.permutation(n)[: n - n % self.batch_size]) if self.shuffle else keep[: n - n % self.batch_size]
        key = self.atoms[keep] * 4096 + self.tokens[keep]
        order = keep[np.argsort(key, kind="stable")]
        if self.shuffle:
            span = self.batch_size * self.neighbourhood
            for s in range(0, len(order), span):
                seg = order[s:s + span]
                rng.shuffle(seg)
                order[s:s + span] = seg
        batches = [order[s:s + self.batch_size].tolist() for s in range(0, len(order), self.batch_size)]
        if self.shuffle:
            perm = rng.permutation(len(batches))
            batches = [batches[i] for i in perm]
        if self.world > 1:
            per = len(batches) // self.world
            batches = batches[self.rank:per * self.world:self.world]
        return batches

    def __iter__(self) -> Iterator[List[int]]:
        return iter(self._batches())

    def __len__(self) -> int:
        n = len(self.atoms)
        nb = n // self.batch_size if (self.drop_last or self.world > 1) else (n + self.batch_size - 1) // self.batch_size
        return nb // self.world if self.world > 1 else nb


def epoch_cost(batches, atoms, tokens) -> float:
    atoms, tokens = np.asarray(atoms), np.asarray(tokens)
    return float(sum(padded_cost(atoms[b], tokens[b]) for b in batches))


class DevicePrefetcher:
    """Iterate ``(net_input: dict[str, Tensor], label: Tensor)`` batches with the NEXT batch's host-to-device copies in
    flight on a side HIP stream while the current step runs (collate / transfer off the critical path).

    The reference moves each batch with synchronous ``.cuda()`` calls inside the step loop (tasks/trainer.py:181-183).
    Here every host tensor is staged through a pinned buffer (re-used while the shapes repeat), copied with
    ``non_blocking=True`` on ``self.stream``, and the consumer's stream waits on the copy's event only when it takes the
    batch -- so a step never waits for PCIe unless the loader itself is the bottleneck.  ``narrow`` (default): only what the
    kernels read is copied (``collate.device_payload``: ``src_edge_type`` as int16, no ``src_coord``) -- at 256 molecules of
    130 atoms the int64 edge types alone are 34.6 MB per batch, 8.7 MB as int16; values are unchanged.
    """

    def __init__(self, batches, device, narrow=True, n_edge_types=None, pad_idx=0):
        import torch
        self._torch = torch
        self.narrow, self.n_edge_types, self.pad_idx = narrow, n_edge_types, pad_idx
        self.batches = batches
        self.device = torch.device(device)
        self.stream = torch.cuda.Stream(device=self.device) if self.device.type == "cuda" else None
        self._pinned = {}

    def _stage(self, name, t):
        torch = self._torch
        if self.stream is None or t.device.type != "cpu":
            return t.to(self.device)
        key = (name, tuple(t.shape), t.dtype)
        buf = self._pinned.get(name)
        if buf is None or (tuple(buf.shape), buf.dtype) != key[1:]:
            buf = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            self._pinned[name] = buf
        buf.copy_(t)
        return buf.to(self.device, non_blocking=True)

    def _launch(self, item):
        torch = self._torch
        net_input, label = item
        if self.narrow:
            from .collate import device_payload
            net_input = device_payload(net_input, self.n_edge_types, self.pad_idx)
        if self.stream is None:
            return {k: (v if k in HOST_FIELDS else self._stage(k, v)) for k, v in net_input.items()}, self._stage("__label__", label), None
        with torch.cuda.stream(self.stream):
            dev_in = {k: (v if k in HOST_FIELDS else self._stage(k, v)) for k, v in net_input.items()}
            dev_lab = self._stage("__label__", label)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        return dev_in, dev_lab, ev

    def __iter__(self):
        torch = self._torch
        it = iter(self.batches)
        try:
            nxt = self._launch(next(it))
        except StopIteration:
            return
        while nxt is not None:
            dev_in, dev_lab, ev = nxt
            if ev is not None:
                torch.cuda.current_stream(self.device).wait_event(ev)
                for t in [v for k, v in dev_in.items() if k not in HOST_FIELDS] + [dev_lab]:
                    t.record_stream(torch.cuda.current_stream(self.device))
                # (the staging buffers are rewritten by the next _launch: its host-side copy_ must not race the DMA)
                ev.synchronize()
            try:
                nxt = self._launch(next(it))
            except StopIteration:
                nxt = None
            yield dev_in, dev_lab
