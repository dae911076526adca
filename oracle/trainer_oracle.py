"""CPU oracle for the caller of the hot path: ``Trainer.fit_predict`` / ``Trainer.predict``.

TEST INFRASTRUCTURE ONLY (same rules as ``mmdti_oracle.py``: imported by ``tests/`` only; the product never imports it).

Restates /root/reference/tasks/trainer.py on top of the functional oracle:
  * :142-162  DataLoader(shuffle=True, drop_last=True, batch_size) over the training set, Adam(lr, eps=1e-6),
              HF ``get_linear_schedule_with_warmup`` (num_training_steps = len(loader) * epochs, warm-up = int(steps * ratio));
  * :177-283  step body, non-AMP branch: ``loss = alpha*task + beta*infonce + beta*ct``; backward; optimizer.step();
              scheduler.step()  (no clipping in this branch, :279-281);
  * :288-306  per-epoch FDS pass over the (re-shuffled) training loader in TRAIN mode under no_grad, then
              ``update_last_epoch_stats(epoch)`` / ``update_running_stats(feats, labels, epoch)``;
  * :308-325  validation predict (eval mode, shuffle=False, short last batch kept), first metric decides the best
              checkpoint (utils/metrics.py:230-258: ``<=`` for decreasing metrics, ``>=`` for increasing ones);
  * :326-328  reload of the best checkpoint, final predict.
Pinned by ``tests/golden/g10_trainer_*.npz`` (the reference's own Trainer run on CPU atop the shims of
``tests/golden/ref_shims.py``): per-step losses, batch orders of every loader pass, final predictions, checkpoint.
"""
from __future__ import annotations

import copy
from typing import Callable, Dict, List

import numpy as np
import torch
from torch.utils.data import DataLoader

from . import mmdti_oracle as O


def collate(samples, pad_idx: int, tokenizer) -> tuple:
    """MM_Model.batch_collate_fn (models/mm_model.py:645-682) for the keys DataHub produces."""
    batch = {}
    first = samples[0][0]
    for k in first.keys():
        if k == "src_coord":
            batch[k] = O.pad_coords([torch.tensor(s[0][k]).float() for s in samples], 0.0)
        elif k == "src_edge_type":
            batch[k] = O.pad_2d([torch.tensor(s[0][k]).long() for s in samples], pad_idx)
        elif k == "src_distance":
            batch[k] = O.pad_2d([torch.tensor(s[0][k]).float() for s in samples], 0.0)
        elif k == "src_tokens":
            batch[k] = O.pad_1d_tokens([torch.tensor(s[0][k]).long() for s in samples], pad_idx)
        elif k == "weights":
            batch[k] = torch.tensor([s[0][k] for s in samples])
    if "smile" in first:
        enc = tokenizer([s[0]["smile"] for s in samples], padding=True, truncation=True, return_tensors="pt")
        batch["input_ids"], batch["attention_mask"] = enc["input_ids"], enc["attention_mask"]
    try:
        label = torch.tensor(np.array([s[1] for s in samples]))
    except Exception:
        label = None
    return batch, label


def linear_warmup_lambda(step: int, warmup: int, total: int) -> float:
    """transformers.optimization.get_linear_schedule_with_warmup's lr_lambda."""
    if step < warmup:
        return float(step) / float(max(1, warmup))
    return max(0.0, float(total - step) / float(max(1, total - warmup)))


def fit_predict(P: Dict[str, torch.Tensor], cfg: O.ModelCfg, train: List, valid: List, tokenizer, hp: dict,
                fds: O.FDSOracle = None, increasing_metric: bool = False, metric: Callable = None, record: dict = None):
    """Returns (y_pred of the reloaded best checkpoint, best params, fds).  ``hp``: learning_rate, batch_size, epochs,
    warmup_ratio, alpha, beta.  ``record`` (optional dict) receives per-step losses and the batch orders."""
    task = cfg.task
    names = sorted(P)
    params = [P[n].detach().clone().requires_grad_(True) for n in names]
    live = dict(zip(names, params))
    pad_idx = cfg.unimol.pad_idx
    ids = {id(s[0]): i for i, s in enumerate(train)}
    ids.update({id(s[0]): 100 + i for i, s in enumerate(valid)})
    orders = []

    def make_collate(phase):
        def fn(samples):
            orders.append([phase] + [ids[id(s[0])] for s in samples] + [-1] * (hp["batch_size"] - len(samples)))
            return collate(samples, pad_idx, tokenizer)
        return fn

    phase = {"v": 0}
    loader = DataLoader(train, batch_size=hp["batch_size"], shuffle=True, drop_last=True, collate_fn=lambda s: make_collate(phase["v"])(s))
    vloader = DataLoader(valid, batch_size=hp["batch_size"], shuffle=False, collate_fn=make_collate(2))
    total = len(loader) * hp["epochs"]
    warm = int(total * hp["warmup_ratio"])
    opt = torch.optim.Adam(params, lr=hp["learning_rate"], eps=1e-6)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda s: linear_warmup_lambda(s, warm, total))
    steps = {"task": [], "infonce": [], "ct": []}

    def target(y):
        return y.float() if task == "regression" else y.long()

    def predict():
        out, losses = [], []
        with torch.no_grad():
            for b, y in vloader:
                o = O.mm_forward(b, live, cfg, training=False)
                losses.append(float(hp["alpha"] * O.task_loss(o["logits"], target(y), task)))
                lg = o["logits"]
                out.append((lg if task == "regression" else torch.softmax(lg, -1)[:, 1:]).numpy())
        return np.concatenate(out), losses

    best, best_state, best_fds = None, None, None
    for epoch in range(hp["epochs"]):
        phase["v"] = 0
        for b, y in loader:
            opt.zero_grad()
            o = O.mm_forward(b, live, cfg, net_target=target(y), fds=fds, epoch=epoch, training=True)
            loss, tl = O.step_loss(o, target(y), task, hp["alpha"], hp["beta"])
            steps["task"].append(float(tl)); steps["infonce"].append(float(o["infonce"])); steps["ct"].append(float(o["ct"]))
            loss.backward()
            opt.step()
            sched.step()
        if fds is not None and epoch >= fds.start_update:
            phase["v"] = 1
            feats, labs = [], []
            with torch.no_grad():
                for b, y in loader:
                    o = O.mm_forward(b, live, cfg, net_target=target(y), fds=fds, epoch=epoch, training=True)
                    feats.append(o["pooled"]); labs.append(target(y))
            fds.update_last_epoch_stats(epoch)
            fds.update_running_stats(torch.cat(feats), torch.cat(labs), epoch)
        y_pred, vloss = predict()
        y_true = np.concatenate([np.asarray(s[1]).reshape(1, -1) for s in valid])
        score = metric(y_true, y_pred)
        better = best is None or (score >= best if increasing_metric else score <= best)
        if better:
            best = score
            best_state = {n: p.detach().clone() for n, p in live.items()}
            best_fds = copy.deepcopy(fds)
    for n, p in live.items():
        p.data.copy_(best_state[n])
    y_pred, _ = predict()
    if record is not None:
        record.update(steps=steps, orders=orders)
    return y_pred, best_state, best_fds
