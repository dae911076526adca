"""Host-side input pipeline helpers (the step's caller side).

The reference batches molecules in dataset order and right-pads every batch to its longest member
(``MM_Model.batch_collate_fn``, models/mm_model.py:645-682; ``pad_1d_tokens`` / ``pad_2d``, utils/util.py:7-72).  The
step's cost is set by the PADDED shape -- pair tensors grow with N_max^2, attention with L_max^2 -- so batching molecules
of similar size is the largest lever a real dataset offers.  ``LengthBucketBatchSampler`` does that without changing what
a batch is (same collate, same tensors, same semantics for InfoNCE / SupCon: the negatives of a step are its batch).
"""
from __future__ import annotations

from typing import Iterator, List, Sequence

import numpy as np


def padded_cost(atoms: Sequence[int], tokens: Sequence[int]) -> float:
    """Relative cost of one padded batch: pair work ~ B * N_max^2 * 15 layers, tower 2 ~ B * L_max * (const + L_max)."""
    b = len(atoms)
    n, l = max(atoms) + 2, max(tokens)
    return b * (15.0 * n * n + 6.0 * l * (12.0 * 512 / 64 + l) / 8.0)


class LengthBucketBatchSampler:
    """Batch sampler (yields lists of dataset indices) that groups molecules of similar (atom count, SMILES token count).

    * every index appears exactly once per epoch (``drop_last=False``) or is dropped only from a final short batch;
    * ``shuffle``: the ORDER of batches and the membership within a size neighbourhood are reshuffled every epoch from
      ``seed + epoch`` (call ``set_epoch``) -- samples still meet different partners from epoch to epoch, which the
      in-batch contrastive losses need;
    * data parallel: ``rank`` / ``world`` deal whole batches round-robin and trim to a common count, so that every rank
      runs the same number of steps (the gradient all-reduce is collective).
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
        key = self.atoms * 4096 + self.tokens
        order = np.argsort(key, kind="stable")
        if self.shuffle:
            span = self.batch_size * self.neighbourhood
            for s in range(0, n, span):
                seg = order[s:s + span]
                rng.shuffle(seg)
                order[s:s + span] = seg
        batches = [order[s:s + self.batch_size].tolist() for s in range(0, n, self.batch_size)]
        if self.drop_last and batches and len(batches[-1]) < self.batch_size:
            batches.pop()
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
        nb = n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size
        return nb // self.world if self.world > 1 else nb


def epoch_cost(batches, atoms, tokens) -> float:
    atoms, tokens = np.asarray(atoms), np.asarray(tokens)
    return float(sum(padded_cost(atoms[b], tokens[b]) for b in batches))
