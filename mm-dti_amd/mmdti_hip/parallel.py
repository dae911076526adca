"""Data parallelism for the fine-tune step: one process per GPU, torch.distributed over RCCL/xGMI.

The reference is single-device; the semantics below are this build's, chosen so that N ranks on a
global batch reproduce the single-process loss and gradients:

  * molecules are split across ranks (weak scaling: B_loc fixed); all weights are replicated;
  * InfoNCE negatives are GLOBAL: each rank contributes its [B_loc, 2*50] pooled projections through ONE all-gather
    (both towers fused into a single latency-bound message); the adjoint is a sum-reduce of the [B_glob, 100] gradient
    from which each rank keeps its rows (all-reduce + slice: 200 KB at 8 x 256, cheaper than a reduce-scatter setup);
  * ConR / SupCon and the task loss stay rank-local means; the step loss is the mean over ranks, so gradients are
    all-reduced with op=AVG, except InfoNCE whose value is already this rank's share of the global loss (hence its
    gradient is pre-multiplied by world_size before the AVG);
  * gradients live in ONE flat fp32 arena (runtime.ParamArena): the all-reduce walks it in large buckets on a side HIP
    stream, and is launched bucket by bucket while the backward of earlier layers is still running (a bucket goes out
    as soon as every parameter in it has reported its gradients enqueued -- ArenaReducer.on_grads_ready).  xGMI is point-to-point (7 links x ~153 GB/s per GPU): a
    ring all-reduce is bound by one link, so few large buckets (default 32 MiB) amortise the per-collective latency.
"""
from __future__ import annotations

import os
from typing import Optional

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None, force: bool = False) -> tuple:
    """torchrun-style env (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT) -> (rank, local_rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # Rehearsal of the multi-rank control flow on a ONE-GPU box (MMDTI_DIST_BACKEND=gloo MMDTI_ONE_GPU=1): every rank on device 0, the
    # collectives through gloo (RCCL refuses two ranks on one device).  Not a performance configuration.
    backend = backend or os.environ.get("MMDTI_DIST_BACKEND") or None
    if os.environ.get("MMDTI_ONE_GPU") == "1":
        local = 0
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
        host_group()        # (created HERE, where every rank is known to be: new_group is collective over the whole world -- ADVICE r03)
    return rank, local, world


def _collectives_active(world: int) -> bool:
    """Collectives run when there is more than one rank -- or, for single-GPU rehearsal of the RCCL code path, when
    MMDTI_FORCE_DDP=1 and a (1-rank) process group exists."""
    return dist.is_initialized() and (world > 1 or os.environ.get("MMDTI_FORCE_DDP") == "1")


class GlobalNegatives:
    """all-gather of the pooled InfoNCE projections and its adjoint (see models/infonce.py:set_global_negatives)."""

    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.active = _collectives_active(self.world)
        self._checked = False

    def _check_equal_rows(self, x: torch.Tensor):
        """all_gather_into_tensor / the adjoint's row slicing assume the same B_loc on every rank; a mismatch would hang
        or slice the wrong rows inside RCCL.  Checked on the first call (every rank makes it at the same point) and on
        every call with MMDTI_DDP_CHECK=1: one 2-element MIN/MAX message."""
        n = torch.tensor([x.shape[0], -x.shape[0]], device=x.device, dtype=torch.int64)
        dist.all_reduce(n, op=dist.ReduceOp.MAX, group=self.group)
        hi, lo = int(n[0]), -int(n[1])
        if hi != lo:
            raise RuntimeError(f"GlobalNegatives: ranks hold different local batch sizes at this step ({lo}..{hi}); use a "
                               "sampler that deals equal batches (data.LengthBucketBatchSampler(world>1), drop_last)")

    def gather(self, x: torch.Tensor) -> torch.Tensor:
        """[B_loc, C] -> [world*B_loc, C] in rank order (equal B_loc on every rank)."""
        if not self.active:
            return x
        x = x.contiguous()
        if not self._checked or os.environ.get("MMDTI_DDP_CHECK") == "1":
            self._check_equal_rows(x)
            self._checked = True
        out = torch.empty(self.world * x.shape[0], x.shape[1], device=x.device, dtype=x.dtype)
        dist.all_gather_into_tensor(out, x, group=self.group)
        return out

    def reduce_scatter(self, g: torch.Tensor) -> torch.Tensor:
        """adjoint of gather: sum over ranks of [world*B_loc, C], keep own rows."""
        if not self.active:
            return g
        g = g.contiguous()
        dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group)
        b = g.shape[0] // self.world
        return g[self.rank * b:(self.rank + 1) * b]

    def row0(self, b_loc: int) -> int:
        return self.rank * b_loc


class ArenaReducer:
    """Bucketed gradient all-reduce (mean) over the flat gradient arena on a side stream.

    Overlap with backward: every module backward of functional.py calls runtime.notify_grads_ready(params) once all
    gradient kernels of `params` are enqueued; on_grads_ready() records a HIP event on that stream (the two towers run
    their backward on different streams) and, as soon as every parameter overlapping a bucket has reported, launches that
    bucket's all-reduce on the communication stream behind those events.  Buckets holding a parameter that never reports
    (no gradient this step) are reduced by finish().  All ranks run the same graph, so the launch order is identical
    everywhere."""

    def __init__(self, arena, bucket_bytes: int = 32 << 20, group=None):
        self.arena, self.group = arena, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.active = _collectives_active(self.world)
        n = arena.numel
        per = max(1, bucket_bytes // 4)
        self.per = per
        self.buckets = [(s, min(n, s + per)) for s in range(0, n, per)]
        self.stream = torch.cuda.Stream() if (self.active and arena.grad.is_cuda) else None
        self.algo = os.environ.get("MMDTI_REDUCE", "ring")
        self._pending = []
        # parameter -> arena range, bucket -> parameters it needs (only when the arena knows its parameters)
        self._range = {}
        self._need = [set() for _ in self.buckets]
        for p in getattr(arena, "params", ()):
            lo = arena.offsets[id(p)]
            hi = lo + p.numel()
            self._range[id(p)] = (lo, hi)
            for b in range(lo // per, (hi - 1) // per + 1):
                self._need[b].add(id(p))
        self._have = [set() for _ in self.buckets]
        self._events = [[] for _ in self.buckets]
        self.overlapped = 0                    # buckets launched from inside backward during the last step (diagnostic)

    def reduce_range(self, lo: int, hi: int):
        """Launch the all-reduce of every not-yet-reduced bucket fully inside [lo, hi) -- for callers that know which
        part of the arena the backward has finished writing."""
        if not self.active:
            return
        for (s, e) in self.buckets:
            if s >= lo and e <= hi and (s, e) not in self._pending:
                self._launch(s, e)

    def on_grads_ready(self, params):
        """runtime.notify_grads_ready hook: `params` have all their gradient kernels enqueued on the current stream."""
        if not self.active:
            return
        ev = None
        if self.stream is not None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
        touched = set()
        for p in params:
            r = self._range.get(id(p))
            if r is None:
                continue
            for b in range(r[0] // self.per, (r[1] - 1) // self.per + 1):
                self._have[b].add(id(p))
                touched.add(b)
        for b in sorted(touched, reverse=True):
            if ev is not None:
                self._events[b].append(ev)
            if self.buckets[b] not in self._pending and self._have[b] >= self._need[b]:
                self._launch(*self.buckets[b], events=self._events[b])
                self.overlapped += 1

    def unreported(self):
        """ids of parameters that did not report during the last backward, per bucket (diagnostic / tests)."""
        return [need - have for need, have in zip(self._need, self._have)]

    def begin_step(self):
        self._have = [set() for _ in self.buckets]
        self._events = [[] for _ in self.buckets]
        self._pending = []
        self.overlapped = 0

    def _mean_over_ranks(self, view):
        """view <- mean over ranks.  SUM + scale rather than ReduceOp.AVG: identical result, no dependence on the collective library's
        AVG support.  Two algorithms (MMDTI_REDUCE):
          "ring"   (default) one RCCL all-reduce per bucket -- the library picks ring / tree for the topology;
          "direct" reduce-scatter as ONE all-to-all (rank j receives chunk j of every rank over its own link to each of them and
                   sums the world chunks) + all-gather of the reduced chunks: on xGMI's point-to-point mesh (7 links x ~153 GB/s
                   per GPU) every link carries 1/world of the bucket in each phase, where a ring is bound by ONE link carrying
                   2 (world-1)/world of it.  Cannot be timed on a 1-GPU box: opt-in until an 8-GPU run has
                   measured it; the arithmetic is covered by the 2-rank gloo tests."""
        if self.world <= 1:
            dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group)      # (1-rank rehearsal: exercises the collective)
            return
        if self.algo != "direct":
            dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group)
            view.mul_(1.0 / self.world)
            return
        n, w = view.numel(), self.world
        chunk = (n + w - 1) // w
        send = view if n == chunk * w else torch.cat([view, view.new_zeros(chunk * w - n)])
        recv = torch.empty_like(send)
        dist.all_to_all_single(recv, send, group=self.group)                   # recv[j*chunk:(j+1)*chunk] = rank j's copy of MY chunk
        mine = recv.view(w, chunk).sum(dim=0).mul_(1.0 / w)
        dist.all_gather_into_tensor(send, mine, group=self.group)
        if send is not view:
            view.copy_(send[:n])

    def _launch(self, s, e, events=None):
        view = self.arena.grad[s:e]
        if self.stream is not None:
            if events is None:
                self.stream.wait_stream(torch.cuda.current_stream())
            else:
                for ev in events:
                    self.stream.wait_event(ev)
            with torch.cuda.stream(self.stream):
                self._mean_over_ranks(view)
        else:
            self._mean_over_ranks(view)
        self._pending.append((s, e))

    def finish(self):
        """Reduce whatever is left and make the main stream wait for the side stream."""
        if not self.active:
            return
        for (s, e) in self.buckets:
            if (s, e) not in self._pending:
                self._launch(s, e)
        if self.stream is not None:
            torch.cuda.current_stream().wait_stream(self.stream)
        self._pending = []


_HOST_GROUP = None


def host_group():
    """A gloo process group for HOST-side exchanges (a few integers per step, decisions of the epoch loop): a collective on host
    memory costs no device synchronisation -- an RCCL all-reduce of two lengths followed by ``int(t[0])`` stalls the host until the
    GPU queue has drained.  The default group itself when it is gloo (CPU runs).  ``dist.new_group`` is collective over the WHOLE world,
    so the group is created eagerly where the process group is set up (init_from_env; FineTuner / Trainer under distributed=True call
    this in their constructors, which every rank runs) -- a lazy first use inside per-step control flow would hang as soon as one
    rank's path skipped it."""
    global _HOST_GROUP
    if _HOST_GROUP is None:
        _HOST_GROUP = dist.group.WORLD if dist.get_backend() == "gloo" else dist.new_group(backend="gloo")
    return _HOST_GROUP


def _host_capable(group):
    """the group a host-memory collective runs on: the caller's if it can take CPU tensors (gloo), else a clear error"""
    if group is None:
        return host_group()
    if dist.get_backend(group) != "gloo":
        raise ValueError("host-side exchanges (padded lengths, epoch decisions) need a gloo group: pass group=None (the library's own gloo side "
                         f"group) or a gloo group, not a {dist.get_backend(group)} one")
    return group


def host_all_reduce_max(values, group=None):
    """element-wise MAX over ranks of a short list of Python ints, through host memory"""
    t = torch.tensor(list(values), dtype=torch.int64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=_host_capable(group))
    return [int(v) for v in t]


def host_broadcast_ints(values, src=0, group=None):
    """rank ``src``'s list of Python ints on every rank, through host memory"""
    t = torch.tensor(list(values), dtype=torch.int64)
    dist.broadcast(t, src=src, group=_host_capable(group))
    return [int(v) for v in t]


PAD_VALUES = {"src_tokens": 0, "src_edge_type": 0, "src_distance": 0.0, "src_coord": 0.0, "input_ids": 1, "attention_mask": 0}


def pad_to_global_lengths(batch: dict, group=None, pad_values: Optional[dict] = None) -> dict:
    """Right-pad a rank's collated batch to the LARGEST atom / token length any rank holds at this step (one 2-int MAX
    all-reduce).  The reference's InfoNCE head averages its per-token projections over ALL positions including padding
    (infonce.py:32-33), so the loss depends on the padded length: with rank-local collation every rank must use the global
    lengths for N ranks to reproduce the single-process value on the union batch.  pad_values: per-key fill (defaults to the
    reference's: dictionary pad 0 for tokens / edge types, 0.0 for distances, RoBERTa pad 1 for input_ids, 0 for the mask).
    The two lengths are tensor SHAPES -- host integers -- and are exchanged as such (host_group: gloo): no device->host
    synchronisation enters the step.  group: a HOST-capable (gloo) group, default host_group()."""
    if not (dist.is_initialized() and (dist.get_world_size() > 1 or os.environ.get("MMDTI_FORCE_DDP") == "1")):
        return batch
    pv = dict(PAD_VALUES, **(pad_values or {}))
    n_loc = batch["src_tokens"].shape[1]
    l_loc = batch["input_ids"].shape[1] if "input_ids" in batch else 0
    n_glob, l_glob = host_all_reduce_max([n_loc, l_loc], group)
    out = {}
    for k, v in batch.items():
        if k in ("src_tokens",):
            grow = (0, n_glob - n_loc)
        elif k in ("src_distance", "src_edge_type"):
            grow = (0, n_glob - n_loc, 0, n_glob - n_loc)
        elif k == "src_coord":
            grow = (0, 0, 0, n_glob - n_loc)
        elif k in ("input_ids", "attention_mask"):
            grow = (0, l_glob - l_loc)
        else:
            out[k] = v
            continue
        out[k] = torch.nn.functional.pad(v, grow, value=pv[k]) if any(grow) else v
    return out


def gather_features(negs: "GlobalNegatives", feats: torch.Tensor, labels: torch.Tensor):
    """FDS statistics under data parallelism (tasks/trainer.py:288-306 has one process): every rank contributes the pooled
    features and labels of its shard; all ranks end with the same (features, labels) of the whole epoch, so the FDS buffers
    -- which are part of the checkpoint -- stay identical across ranks."""
    if negs is None or negs.world <= 1:
        return feats, labels
    return negs.gather(feats.contiguous()), negs.gather(labels.float().view(labels.shape[0], -1).contiguous())


def shard_batch(batch: dict, label, rank: int, world: int):
    """Contiguous split of an already-collated GLOBAL batch.  Every shard keeps the global padded lengths, which keeps
    the reference's unmasked InfoNCE mean (infonce.py:32-33) identical to the single-process value."""
    B = label.shape[0]
    assert B % world == 0, "global batch must divide evenly (drop_last on the global sampler)"
    b = B // world
    sl = slice(rank * b, (rank + 1) * b)
    return {k: (v[sl] if torch.is_tensor(v) else v) for k, v in batch.items()}, label[sl]      # (host scalars -- token_pad_id, packable -- describe every shard)
