"""The fine-tune step of the reference's ``Trainer.fit_predict`` (tasks/trainer.py:177-283) as an MI355X-native engine.

Same loss composition (``alpha*task + beta*infonce + beta*ct``, :192-193), same optimizer maths (Adam, eps 1e-6, :160),
same linear warm-up schedule (HF ``get_linear_schedule_with_warmup``, :161-162), same gradient clipping (max_norm 5.0,
AMP branch :274), same per-epoch FDS statistics pass (:288-306).  What changes is where the work runs:

  * bf16 MFMA compute with fp32 master weights instead of fp16 autocast + GradScaler (no loss scaling needed);
  * parameters, gradients, Adam moments and the bf16 weight shadow are flat arenas: zero-grad, gradient all-reduce,
    clip + Adam + shadow refresh are each ONE pass over contiguous HBM;
  * no host synchronisation inside the step: the four logged scalars stay on the device (``StepOutput``) and are
    fetched by the caller when it wants them (the reference does four ``float(t.data)`` syncs per step, :195-197,238);
  * optional data parallelism (``parallel.py``): global InfoNCE negatives + bucketed gradient all-reduce over RCCL.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Optional

import torch

from .functional import CELossFn, MSELossFn, BCELogitsLossFn
from .parallel import ArenaReducer, GlobalNegatives, gather_features
from .runtime import ParamArena, add_grad_ready_hook, remove_grad_ready_hook, dropout_state


@dataclass
class StepOutput:
    loss: torch.Tensor                      # this rank's step loss (alpha*task + beta*infonce_global + beta*ct)
    task_loss: torch.Tensor
    infonce_loss: Optional[torch.Tensor]    # the GLOBAL InfoNCE value (under DDP: sum of the ranks' shares)
    ct_loss: Optional[torch.Tensor]
    logits: torch.Tensor


def linear_warmup_lr(base_lr: float, step: int, warmup: int, total: int) -> float:
    """transformers.get_linear_schedule_with_warmup: lambda of the step count AFTER `step` scheduler.step() calls."""
    if step < warmup:
        return base_lr * float(step) / float(max(1, warmup))
    return base_lr * max(0.0, float(total - step) / float(max(1, total - warmup)))


def _qkv_groups(model):
    """query/key/value Linears of every BERT-style attention block: their weights (and biases) go back to back in the
    arena so that the layer can run them as ONE [3D, D] GEMM (functional._bert_layer_fwd)."""
    groups = []
    for m in model.modules():
        q, k, v = (getattr(m, n, None) for n in ("query", "key", "value"))
        if all(isinstance(t, torch.nn.Module) and hasattr(t, "weight") for t in (q, k, v)) and q.weight.shape == k.weight.shape == v.weight.shape:
            groups.append((q.weight, k.weight, v.weight))
            if all(getattr(t, "bias", None) is not None for t in (q, k, v)):
                groups.append((q.bias, k.bias, v.bias))
    return groups


class FineTuner:
    def __init__(self, model, task: str, learning_rate=1e-4, adam_eps=1e-6, warmup_ratio=0.03, total_steps=1000, alpha=1.0, beta=0.1,
                 max_norm: Optional[float] = 5.0, distributed: bool = False, bucket_bytes: int = 32 << 20):
        self.model, self.task = model, task
        self.lr, self.eps, self.alpha, self.beta, self.max_norm = learning_rate, adam_eps, alpha, beta, max_norm
        self.total_steps = total_steps
        self.warmup = int(total_steps * warmup_ratio)
        self.sched_step = 0
        self.arena = ParamArena(model.parameters(), adjacent=_qkv_groups(model))
        self._graphs, self._state, self._salt, self._salted = {}, None, None, False
        self.world = 1
        self.reducer = None
        remove_grad_ready_hook(self)
        if distributed:
            self.negs = GlobalNegatives()
            self.world = self.negs.world
            if self.negs.active:
                from .parallel import host_group
                host_group()            # (the gloo side group of the host-side exchanges: created where every rank is -- ADVICE r03)
                # replicas must start identical: the cross-modal block, InfoNCE head and classification head are random-
                # initialised, so rank 0's arena is broadcast (no reliance on every rank seeding alike) ...
                torch.distributed.broadcast(self.arena.data, src=0)
                self.arena.refresh_shadow()
                # ... while dropout masks must DIFFER across ranks (each rank holds different molecules)
                dropout_state.reseed(dropout_state.base + 0x9E37 * (self.negs.rank + 1))
            self.reducer = ArenaReducer(self.arena, bucket_bytes)
            self._b_loc = None
            # gradient buckets leave during backward (MMDTI_NO_REDUCE_OVERLAP=1: all of them after it)
            if os.environ.get("MMDTI_NO_REDUCE_OVERLAP") != "1":
                add_grad_ready_hook(self, self.reducer.on_grads_ready)
        # built-in task-loss kernels (models/nnmodel.py:24-34): MSE, cross-entropy, and BCE-with-logits for the multilabel table's
        # 'bce' entry; every other task / loss (focal, GHM, MAE-with-NaN ...) runs as the callable the caller passes as `loss_func`
        if task == "regression":
            self.task_loss = lambda lg, y: MSELossFn.apply(lg, y.float())
        elif task in ("classification", "multiclass"):
            self.task_loss = lambda lg, y: CELossFn.apply(lg, y)
        elif task == "multilabel_classification":
            self.task_loss = lambda lg, y: BCELogitsLossFn.apply(lg, y)
        else:
            self.task_loss = None

    # ------------------------------------------------------------------
    def _bind_global_negatives(self, b_loc: int):
        if self.negs.active and self._b_loc != b_loc:
            self.model.infonce.set_global_negatives(self.negs.gather, self.negs.reduce_scatter, self.negs.row0(b_loc))
            self._b_loc = b_loc

    def forward_backward(self, net_input: dict, net_target: torch.Tensor, epoch: int = 0, use_weight: bool = False,
                         return_infonce_loss: bool = True, return_ct_loss: bool = True, loss_func=None) -> StepOutput:
        """optimizer.zero_grad(); model(...); loss; loss.backward()  (+ gradient all-reduce under DDP).  The two flags
        select the reference's four call forms (tasks/trainer.py:184-212); ``loss_func`` replaces the built-in task-loss
        kernel with any callable on (logits, target)."""
        model = self.model
        self.arena.zero_grad()
        if self.reducer is not None:
            self._bind_global_negatives(net_target.shape[0])
            self.reducer.begin_step()
        kw = dict(epoch=epoch)
        if return_infonce_loss:
            kw["return_infonce_loss"] = True
        if return_ct_loss:
            kw.update(return_ct_loss=True, use_weight=use_weight)
        if return_ct_loss or return_infonce_loss:
            kw["net_target"] = net_target
        out = model(**net_input, **kw)
        out = out if isinstance(out, tuple) else (out,)
        logits = out[0]
        infonce = out[1] if return_infonce_loss else None
        ct = out[-1] if (return_ct_loss and len(out) > (2 if return_infonce_loss else 1)) else None
        lf = loss_func or self.task_loss
        if lf is None:
            raise ValueError(f"FineTuner: task {self.task!r} has no built-in loss kernel -- pass the task loss as `loss_func` "
                             "(any callable on (logits, target), e.g. the reference's LOSS_RREGISTER entry, models/nnmodel.py:24-34)")
        tl = lf(logits, net_target)
        loss = self.alpha * tl
        if infonce is not None:
            # under DDP `infonce` is this rank's share of the GLOBAL loss: x world so that the rank-mean of gradients is exact
            loss = loss + self.beta * (infonce * self.world if self.world > 1 else infonce)
        if ct is not None:
            loss = loss + self.beta * ct
        loss.backward()
        infonce_global = None if infonce is None else infonce.detach()
        if self.reducer is not None:
            if infonce is not None and self.negs.active and self.world > 1:
                infonce_global = infonce_global.clone()
                torch.distributed.all_reduce(infonce_global, op=torch.distributed.ReduceOp.SUM)      # 4 bytes, for logging
            self.reducer.finish()
        return StepOutput(loss.detach(), tl.detach(), infonce_global, None if ct is None else ct.detach(), logits.detach())

    def optimizer_step(self):
        lr = linear_warmup_lr(self.lr, self.sched_step, self.warmup, self.total_steps)
        self.arena.adam_step(lr, eps=self.eps, max_norm=self.max_norm)
        self.sched_step += 1

    # ------------------------------------------------------------------ the whole step as ONE HIP graph
    def graphed_step(self, net_input: dict, net_target: torch.Tensor, epoch: int = 0, use_weight: bool = False, **kw) -> StepOutput:
        """step() replayed from a captured HIP graph: ~900 kernel launches become one graph launch, which is what the step
        costs at the reference's real batch sizes (16-32: launch-bound, DESIGN.md "small batches").  One graph per input
        shape signature (keep the number of distinct padded shapes small: bucketed batches, lengths rounded up).
        What changes from step to step lives on the device: the optimizer-step counter, the learning rate of the HF schedule,
        Adam's bias corrections and the dropout salt (ops.step_state_advance is the first node of the graph); the inputs
        are copied into the graph's static buffers before each replay.  Returned tensors are the graph's static outputs --
        read them before the next call.  Not available under data parallelism (collectives are left out of graphs here)."""
        if self.reducer is not None:
            raise RuntimeError("graphed_step: not supported with distributed=True (use step())")
        # Host-side batch descriptors (atom_counts, token_counts, ...) select kernels and tile counts on the HOST at capture time;
        # a replay with another batch of the same padded shape would run with the captured batch's lengths.  The graph therefore
        # runs the padded layout: correct for any batch of the shape.
        from .collate import HOST_FIELDS
        net_input = {k: v for k, v in net_input.items() if k not in HOST_FIELDS}
        key = (epoch >= getattr(getattr(self.model, "fds_cfg", None), "start_smooth", 1 << 30), bool(use_weight), tuple(sorted(kw.items())),
               tuple((k, tuple(v.shape), v.dtype) for k, v in sorted(net_input.items())), tuple(net_target.shape), net_target.dtype)
        ent = self._graphs.get(key)
        if ent is None:
            ent = self._capture(net_input, net_target, epoch, use_weight, kw)
            self._graphs[key] = ent
        static_in, static_tgt, graph, out = ent
        for k, v in net_input.items():
            static_in[k].copy_(v, non_blocking=True)
        static_tgt.copy_(net_target, non_blocking=True)
        # the device-resident step counter follows the host's: eager steps taken since the last replay (or another graph's replays)
        # have advanced the schedule too -- one 4-byte fill, enqueued
        self._state[0:1].fill_(float(self.sched_step))
        self._salted = True
        graph.replay()
        self.sched_step += 1
        self.arena.step_count += 1
        return out

    def _state_step(self, net_input, net_target, epoch, use_weight, kw):
        """The body that is captured: advance the device state, then forward / backward / clip + Adam reading it."""
        from . import ops
        ops.step_state_advance(self._state, self._salt, self.lr, self.warmup, self.total_steps)
        out = self.forward_backward(net_input, net_target, epoch, use_weight, **kw)
        self.arena.adam_step(0.0, eps=self.eps, max_norm=self.max_norm, step_state=self._state)
        self.arena.step_count -= 1                   # (the host-side counter is advanced by graphed_step, once per replay)
        return out

    def _capture(self, net_input, net_target, epoch, use_weight, kw):
        dev = net_target.device
        if self._state is None:
            self._state = torch.zeros(4, device=dev, dtype=torch.float32)
            self._salt = torch.zeros(2, device=dev, dtype=torch.int64)
            self._salt[0] = dropout_state.base & 0x7FFFFFFFFFFFFFFF
        # the device counter continues from wherever the eager path (or another graph) left the schedule
        self._state[0] = float(self.sched_step)
        if self.arena.adam_m is None:
            self.arena.adam_m = torch.zeros_like(self.arena.data)
            self.arena.adam_v = torch.zeros_like(self.arena.data)
        static_in = {k: v.clone() for k, v in net_input.items()}
        static_tgt = net_target.clone()
        # warm-up on a side stream (lazy stream / buffer creation must not happen inside the capture), with a throw-away copy of
        # the state this step mutates
        saved = (self.arena.data.clone(), self.arena.adam_m.clone(), self.arena.adam_v.clone(), self._state.clone(), self._salt.clone(),
                 self.arena.step_count)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):
                self._state_step(static_in, static_tgt, epoch, use_weight, kw)
        torch.cuda.current_stream().wait_stream(s)
        self.arena.data.copy_(saved[0]); self.arena.adam_m.copy_(saved[1]); self.arena.adam_v.copy_(saved[2])
        self._state.copy_(saved[3]); self._salt.copy_(saved[4])
        self.arena.step_count = saved[5]
        self.arena.refresh_shadow()
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = self._state_step(static_in, static_tgt, epoch, use_weight, kw)
        return static_in, static_tgt, graph, out

    def step(self, net_input: dict, net_target: torch.Tensor, epoch: int = 0, use_weight: bool = False, **kw) -> StepOutput:
        if getattr(self, "_salted", False):
            # graph replays left their dropout salt in the kernel libraries: an eager step draws its masks from the by-value
            # (seed, site) pairs alone, as every other engine in the process expects (enqueued, no synchronisation)
            from . import ops
            ops.seed_salt_reset(sync=False)
            self._salted = False
        out = self.forward_backward(net_input, net_target, epoch, use_weight, **kw)
        self.optimizer_step()
        return out

    # ------------------------------------------------------------------ tasks/trainer.py:288-306
    @torch.no_grad()
    def fds_epoch_pass(self, batches, epoch: int):
        """Full pass over the training batches (model left in train mode as in the reference) collecting pooled features,
        then FDS.update_last_epoch_stats / update_running_stats.  Under DDP the features are all-gathered so every rank
        ends with identical buffers."""
        model = self.model
        feats, labels = [], []
        for net_input, net_target in batches:
            _, f = model(**net_input, epoch=epoch, return_feature=True, net_target=net_target)
            feats.append(f)
            labels.append(net_target)
        f, y = torch.cat(feats), torch.cat(labels)
        if self.world > 1:
            f, y = gather_features(self.negs, f, y)
        model.FDS.update_last_epoch_stats(epoch)
        model.FDS.update_running_stats(f, y, epoch)
