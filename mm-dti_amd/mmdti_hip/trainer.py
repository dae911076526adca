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

from .functional import CELossFn, MSELossFn
from .parallel import ArenaReducer, GlobalNegatives
from .runtime import ParamArena, set_grad_ready_hook


@dataclass
class StepOutput:
    loss: torch.Tensor
    task_loss: torch.Tensor
    infonce_loss: Optional[torch.Tensor]
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
        self.world = 1
        self.reducer = None
        set_grad_ready_hook(None)
        if distributed:
            self.negs = GlobalNegatives()
            self.world = self.negs.world
            self.reducer = ArenaReducer(self.arena, bucket_bytes)
            self._b_loc = None
            # gradient buckets leave during backward (MMDTI_NO_REDUCE_OVERLAP=1: all of them after it); sub-batched
            # tower 1 accumulates into the same gradients several times per step, so it keeps the after-backward form
            if os.environ.get("MMDTI_NO_REDUCE_OVERLAP") != "1" and getattr(model, "split_tower1", 1) == 1:
                set_grad_ready_hook(self.reducer.on_grads_ready)
        if task == "regression":
            self.task_loss = lambda lg, y: MSELossFn.apply(lg, y.float())
        elif task in ("classification", "multiclass"):
            self.task_loss = lambda lg, y: CELossFn.apply(lg, y)
        else:
            raise NotImplementedError(f"task loss for {task!r} is outside the hot path of this build (models/nnmodel.py:24-34)")

    # ------------------------------------------------------------------
    def _bind_global_negatives(self, b_loc: int):
        if self.negs.active and self._b_loc != b_loc:
            self.model.infonce.set_global_negatives(self.negs.gather, self.negs.reduce_scatter, self.negs.row0(b_loc))
            self._b_loc = b_loc

    def forward_backward(self, net_input: dict, net_target: torch.Tensor, epoch: int = 0, use_weight: bool = False) -> StepOutput:
        """optimizer.zero_grad(); model(...); loss; loss.backward()  (+ gradient all-reduce under DDP)."""
        model = self.model
        self.arena.zero_grad()
        if self.reducer is not None:
            self._bind_global_negatives(net_target.shape[0])
            self.reducer.begin_step()
        logits, infonce, ct = model(**net_input, return_infonce_loss=True, return_ct_loss=True, net_target=net_target, use_weight=use_weight,
                                    epoch=epoch)
        tl = self.task_loss(logits, net_target)
        # under DDP `infonce` is this rank's share of the GLOBAL loss: x world so that the rank-mean of gradients is exact
        loss = self.alpha * tl + self.beta * (infonce * self.world if self.world > 1 else infonce) + self.beta * ct
        loss.backward()
        if self.reducer is not None:
            self.reducer.finish()
        return StepOutput(loss.detach(), tl.detach(), infonce.detach(), ct.detach(), logits.detach())

    def optimizer_step(self):
        lr = linear_warmup_lr(self.lr, self.sched_step, self.warmup, self.total_steps)
        self.arena.adam_step(lr, eps=self.eps, max_norm=self.max_norm)
        self.sched_step += 1

    def step(self, net_input: dict, net_target: torch.Tensor, epoch: int = 0, use_weight: bool = False) -> StepOutput:
        out = self.forward_backward(net_input, net_target, epoch, use_weight)
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
            f, y = self.negs.gather(f), self.negs.gather(y.float().view(y.shape[0], -1))
        model.FDS.update_last_epoch_stats(epoch)
        model.FDS.update_running_stats(f, y, epoch)
