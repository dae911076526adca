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
    """Iterate ``(net_input: dict[str, Tensor], label: Tensor)`` batches with the NEXT batch's host-to-device copies in flight on a
    side HIP stream while the current step runs (collate / transfer off the critical path).

    The reference moves each batch with synchronous ``.cuda()`` calls inside the step loop (tasks/trainer.py:181-183) and collates
    in the main process (num_workers = 0, :551-555).  Here a batch is handed to the consumer FIRST -- which enqueues its step,
    asynchronously -- and only then is the next one pulled from the wrapped iterable, narrowed (``collate.device_payload``: int16 edge
    types, no ``src_coord``, packing facts; values unchanged -- at 256 molecules of 130 atoms the int64 edge types alone are 34.6 MB
    per batch, 8.7 MB as int16), staged and copied with ``non_blocking=True`` on ``self.stream``; the consumer's stream waits for the
    copy event when it takes the batch.  Staging: tensors that arrive PINNED (a ``DataLoader(pin_memory=True)`` whose worker processes
    ran ``collate.HostCollate`` -- the production form, ``tasks.Trainer(num_workers=k)``) are copied from where they lie; pageable ones
    go through a ring of ``depth + 1`` pinned buffer sets, a set being rewritten only after the copy that last read it has completed.

    Round 4 (bench.py ``workloads.pipeline``): with ONE staging set the loop waited 15 ms per step for the previous copy (the blit
    shares the chip with the step's kernels) before it could stage the next batch, and it prepared batch i + 1 BEFORE handing over batch
    i, so the GPU idled through every collate: 87 ms per mixed-length step for 31 ms of kernels.  ``threaded=True`` moves payload and
    staging to a Python thread -- which only pays when the wrapped iterable's own work releases the GIL (measured with a Python-loop
    collate: 113 ms per step, the two threads fighting over the interpreter); worker PROCESSES are what takes collate off the
    critical path.
    """

    def __init__(self, batches, device, narrow=True, n_edge_types=None, pad_idx=0, depth=2, threaded=False):
        import torch
        self._torch = torch
        self.narrow, self.n_edge_types, self.pad_idx = narrow, n_edge_types, pad_idx
        self.batches = batches
        self.device = torch.device(device)
        self.stream = torch.cuda.Stream(device=self.device) if self.device.type == "cuda" else None
        self.depth = max(1, int(depth))
        self.threaded = bool(threaded) and self.stream is not None
        self._ring = [dict(buf={}, event=None) for _ in range(self.depth + 1)]

    # ---- host side (background thread): payload + pinned staging
    def _stage_host(self, slot, item):
        """-> (staged: {name: pinned tensor | host field}, label pinned tensor).  Runs on the staging thread."""
        torch = self._torch
        net_input, label = item
        if self.narrow:
            from .collate import device_payload
            net_input = device_payload(net_input, self.n_edge_types, self.pad_idx)
        def pin(name, t):
            if not torch.is_tensor(t) or t.device.type != "cpu" or self.stream is None or t.is_pinned():
                return t                       # (pinned already -- a pin_memory DataLoader: copied from where it lies, torch's host allocator keeps it alive)
            if slot["event"] is not None:      # the copy that last read this set of staging buffers must be done before they are rewritten
                slot["event"].synchronize()
                slot["event"] = None
            # (one grow-only pinned byte buffer per field and set: the padded length changes from batch to batch, and a pinned
            #  allocation per new shape -- page-locking 26 MB -- costs more than the copy it serves)
            need = t.numel() * t.element_size()
            raw = slot["buf"].get(name)
            if raw is None or raw.numel() < need:
                raw = torch.empty(max(256, need + need // 4), dtype=torch.uint8, pin_memory=True)
                slot["buf"][name] = raw
            buf = raw[:need].view(t.dtype).view(t.shape)
            buf.copy_(t)
            return buf

        staged = {k: (v if k in HOST_FIELDS else pin(k, v)) for k, v in net_input.items()}
        return staged, pin("__label__", label)

    # ---- device side (calling thread): enqueue the copies
    def _enqueue(self, slot, staged, label):
        torch = self._torch
        if self.stream is None:
            dev_in = {k: (v if k in HOST_FIELDS else v.to(self.device)) for k, v in staged.items()}
            return dev_in, label.to(self.device), None
        with torch.cuda.stream(self.stream):
            dev_in = {k: (v if k in HOST_FIELDS else v.to(self.device, non_blocking=True)) for k, v in staged.items()}
            dev_lab = label.to(self.device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        slot["event"] = ev
        return dev_in, dev_lab, ev

    def _launch(self, item, slot=None):
        """stage + enqueue on the calling thread (the unthreaded path; bench.py wraps it to time one batch's transfer)"""
        slot = self._ring[0] if slot is None else slot
        staged, label = self._stage_host(slot, item)
        return self._enqueue(slot, staged, label)

    def _hand_over(self, dev_in, dev_lab, ev):
        torch = self._torch
        if ev is not None:
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(ev)
            for t in [v for k, v in dev_in.items() if k not in HOST_FIELDS] + [dev_lab]:
                t.record_stream(cur)
        return dev_in, dev_lab

    def __iter__(self):
        if not self.threaded:
            yield from self._iter_inline()
            return
        import queue
        import threading
        q = queue.Queue(maxsize=self.depth)
        stop = threading.Event()

        def produce():
            try:
                k = 0
                for item in self.batches:
                    if stop.is_set():
                        return
                    slot = self._ring[k % len(self._ring)]
                    staged, label = self._stage_host(slot, item)
                    q.put((slot, staged, label))
                    k += 1
                q.put(None)
            except BaseException as e:      # noqa: BLE001  (handed to the consumer, re-raised there)
                q.put(e)

        th = threading.Thread(target=produce, name="mmdti-stage", daemon=True)
        th.start()
        try:
            nxt = q.get()
            pending = None
            while True:
                if isinstance(nxt, BaseException):
                    raise nxt
                if nxt is not None:
                    slot, staged, label = nxt
                    ready = self._enqueue(slot, staged, label)       # copies of batch i + 1 in flight ...
                else:
                    ready = None
                if pending is not None:
                    yield self._hand_over(*pending)                  # ... while the consumer enqueues the step on batch i
                if ready is None:
                    break
                pending = ready
                nxt = q.get()
        finally:
            stop.set()
            while th.is_alive():            # unblock a producer waiting on a full queue
                try:
                    q.get_nowait()
                except queue.Empty:
                    th.join(timeout=0.01)

    def _iter_inline(self):
        it = iter(self.batches)
        k = 0
        try:
            nxt = self._launch(next(it), self._ring[0])
        except StopIteration:
            return
        while nxt is not None:
            cur = self._hand_over(*nxt)
            k += 1
            # hand the batch over FIRST: the consumer enqueues its step (asynchronously), and only then does this generator resume to
            # collate / stage the next batch -- on the host, while the GPU runs the step just enqueued
            yield cur
            try:
                nxt = self._launch(next(it), self._ring[k % len(self._ring)])
            except StopIteration:
                nxt = None
