"""Drop-in for the reference's ``tasks/trainer.py`` -- the caller of the hot path.

``Trainer`` keeps the reference's constructor keywords (/root/reference/tasks/trainer.py:29-70), ``fit_predict`` /
``predict`` signatures (:122, :387), ``decorate_batch`` / ``decorate_torch_batch`` (:72-120), ``set_seed`` (:529-537) and
``NNDataLoader`` (:540-556), so ``models/nnmodel.py`` (``NNModel.run`` / ``evaluate``) drives it unchanged.  Behaviour
reproduced, with the reference's own run frozen in tests/golden/g10_trainer_*.npz as the check:

  * the same ``DataLoader(shuffle=True, drop_last=True)`` over the training set and ``shuffle=False`` loaders for
    validation, so batch membership per step is the reference's for the same torch seed;
  * Adam(lr, eps=1e-6), HF linear warm-up/decay over ``len(loader) * epochs`` steps, warm-up ``int(steps*warmup_ratio)``;
  * ``loss = alpha*task + beta*infonce + beta*ct`` in the four call forms selected by return_infonce_loss/return_ct_loss;
  * gradient clipping at ``max_norm`` only in the AMP branch (``use_amp``), none otherwise (:270-281);
  * the per-epoch FDS pass over the re-shuffled training loader in train mode under no_grad (:288-306);
  * validation each epoch, first metric decides; best checkpoint ``{'model_state_dict': ...}`` written to
    ``dump_dir/model_{fold}.pth`` (rank 0 only under data parallelism), early stopping with ``patience``, reload of the
    best checkpoint and the final prediction (:308-328).

What changes is where the work runs: the step goes through ``mmdti_hip.trainer.FineTuner`` (bf16 MFMA kernels with fp32
master weights in one flat arena, fused clip+Adam, no GradScaler -- bf16 needs no loss scaling); the four logged scalars
stay on the device and are fetched once per epoch instead of four ``float()`` synchronisations per step (:195-197,238);
FDS features never visit the host (:302-304 round-trips them through numpy and hard-codes ``.cuda()``).
There is no CPU path: ``use_cuda=False`` raises.
"""
from __future__ import annotations

import os
import time
import logging

import numpy as np
import torch
from torch.utils.data import DataLoader as TorchDataLoader

from ..trainer import FineTuner
from ..collate import HostCollate, device_payload, to_device

logger = logging.getLogger("mmdti_hip")


# ------------------------------------------------------------------------------------------------ epoch metric
def _auc(y, p):
    """ROC AUC by the rank statistic (ties get mid-ranks), one column."""
    y, p = np.asarray(y).ravel().astype(int), np.asarray(p).ravel().astype(np.float64)
    order = np.argsort(p, kind="mergesort")
    ranks = np.empty(len(p))
    sp = p[order]
    i = 0
    while i < len(sp):
        j = i
        while j + 1 < len(sp) and sp[j + 1] == sp[i]:
            j += 1
        ranks[order[i:j + 1]] = 0.5 * (i + j) + 1.0
        i = j + 1
    npos, nneg = int(y.sum()), int(len(y) - y.sum())
    if npos == 0 or nneg == 0:
        raise ValueError("AUC needs both classes in the validation set")
    return float((ranks[y == 1].sum() - npos * (npos + 1) / 2.0) / (npos * nneg))


def _log_loss(y, p):
    """binary log loss of P(class 1), probabilities clipped to [eps, 1 - eps] (sklearn.metrics.log_loss).  eps follows the dtype the
    reference hands sklearn -- ``predict.astype(np.float32)`` (utils/metrics.py:168), so float32's eps: on saturated softmax outputs the
    float64 eps gives another number (y = [0,1,1,0,1], p = [1,1,.7,.2,0]: 6.49 vs 14.53 -- ADVICE r03)."""
    y, p = np.asarray(y).ravel().astype(np.float64), np.asarray(p).ravel().astype(np.float32)
    eps = np.finfo(np.float32).eps
    p = np.clip(p.astype(np.float64), eps, 1.0 - eps)
    return float(-np.mean(y * np.log(p) + (1.0 - y) * np.log(1.0 - p)))


def _auprc(y, p):
    """average precision: sum over the distinct score thresholds (descending) of (recall step) x precision"""
    y, p = np.asarray(y).ravel().astype(int), np.asarray(p).ravel().astype(np.float64)
    if y.sum() == 0:
        raise ValueError("average precision needs a positive in the validation set")
    order = np.argsort(-p, kind="mergesort")
    ys, ps = y[order], p[order]
    last = np.r_[np.nonzero(np.diff(ps))[0], len(ps) - 1]          # last index of every group of tied scores
    tp = np.cumsum(ys)[last]
    precision, recall = tp / (last + 1.0), tp / float(y.sum())
    return float(np.sum(np.diff(np.r_[0.0, recall]) * precision))


def _confusion(y, p):
    y, q = np.asarray(y).ravel().astype(int), (np.asarray(p).ravel() > 0.5).astype(int)
    return float(((q == 1) & (y == 1)).sum()), float(((q == 1) & (y == 0)).sum()), float(((q == 0) & (y == 1)).sum()), float(((q == 0) & (y == 0)).sum())


def _f1(y, p):
    tp, fp, fn, _ = _confusion(y, p)
    return 0.0 if tp == 0 else float(2 * tp / (2 * tp + fp + fn))


def _mcc(y, p):
    tp, fp, fn, tn = _confusion(y, p)
    den = np.sqrt((tp + fp) * (tp + fn) * (tn + fp) * (tn + fn))
    return 0.0 if den == 0 else float((tp * tn - fp * fn) / den)


def _pearson(y, p):
    y, p = np.asarray(y).ravel().astype(np.float64), np.asarray(p).ravel().astype(np.float64)
    return float(np.corrcoef(y, p)[0, 1])


def _midranks(v):
    order = np.argsort(v, kind="mergesort")
    r = np.empty(len(v))
    sv, i = v[order], 0
    while i < len(sv):
        j = i
        while j + 1 < len(sv) and sv[j + 1] == sv[i]:
            j += 1
        r[order[i:j + 1]] = 0.5 * (i + j) + 1.0
        i = j + 1
    return r


_METRIC_TABLE = {      # name -> (function(y_true, y_pred), larger_is_better)   (directions as utils/metrics.py:72-112)
    "mse": (lambda y, p: float(np.mean((y - p) ** 2)), False),
    "rmse": (lambda y, p: float(np.sqrt(np.mean((y - p) ** 2))), False),
    "mae": (lambda y, p: float(np.mean(np.abs(y - p))), False),
    "r2": (lambda y, p: float(1.0 - np.sum((y - p) ** 2) / np.sum((y - np.mean(y)) ** 2)), True),
    "pearsonr": (_pearson, True),
    "spearmanr": (lambda y, p: _pearson(_midranks(np.asarray(y).ravel().astype(np.float64)), _midranks(np.asarray(p).ravel().astype(np.float64))), True),
    "auc": (_auc, True),
    "auroc": (_auc, True),
    "auprc": (_auprc, True),
    "log_loss": (_log_loss, False),
    "acc": (lambda y, p: float(np.mean((np.asarray(p) > 0.5).astype(int) == np.asarray(y).astype(int))), True),
    "f1_score": (_f1, True),
    "mcc": (_mcc, True),
}
# the FIRST default metric of every task is the reference's (utils/metrics.py:114-120 DEFAULT_METRICS): it is the one logged and, with
# metrics other than 'loss' / 'none' / '', the one that drives early stopping
_DEFAULT_METRIC = {"regression": "mse", "classification": "log_loss", "multilabel_classification": "log_loss", "multilabel_regression": "mse"}


class EpochMetric:
    """The slice of ``utils.Metrics`` the trainer needs: ``cal_metric`` -> ordered dict whose FIRST entry drives early
    stopping, and that entry's direction.  Inside the reference's tree pass its own ``utils.Metrics`` object through the
    ``metrics_obj`` keyword instead (the full sklearn/scipy table lives there and is outside the hot path)."""

    def __init__(self, task, metrics_str):
        names = [m for m in (metrics_str or "").split(",") if m] if isinstance(metrics_str, str) and metrics_str not in ("none", "loss") else []
        if not names:
            names = [_DEFAULT_METRIC.get(task, "mse")]
        for n in names:
            if n not in _METRIC_TABLE:
                raise ValueError('Unknown metric: {}'.format(n))
        self.names = names

    def cal_metric(self, label, predict, nan_value=-1.0, threshold=0.5, label_cnt=None):
        label, predict = np.asarray(label, dtype=np.float64), np.asarray(predict, dtype=np.float64)
        return {n: float(np.mean([_METRIC_TABLE[n][0](label[:, c], predict[:, c]) for c in range(label.shape[1])])) for n in self.names}

    def is_increase(self, name):
        return _METRIC_TABLE[name][1]


def _is_increase(metrics, task, name):
    if isinstance(metrics, EpochMetric):
        return metrics.is_increase(name)
    return bool(metrics.METRICS_REGISTER[name][1])          # the reference's utils.Metrics


# ------------------------------------------------------------------------------------------------ the trainer
class Trainer(object):
    def __init__(self, save_path=None, **params):
        self.save_path = save_path
        self.task = params.get('task', None)
        if self.task != 'repr':
            self.metrics_str = params['metrics']
            self.metrics = params.get('metrics_obj') or EpochMetric(self.task, self.metrics_str)
        self._init_trainer(**params)

    def _init_trainer(self, **params):
        self.seed = params.get('seed', 42)
        self.set_seed(self.seed)
        self.logger_level = int(params.get('logger_level', 1))
        self.learning_rate = float(params.get('learning_rate', 1e-4))
        self.batch_size = params.get('batch_size', 32)
        self.max_epochs = params.get('epochs', 50)
        self.warmup_ratio = params.get('warmup_ratio', 0.1)
        self.patience = params.get('patience', 10)
        self.max_norm = params.get('max_norm', 1.0)
        self.cuda = params.get('use_cuda', False)
        self.amp = params.get('use_amp', False)
        self.device = torch.device("cuda" if torch.cuda.is_available() and self.cuda else "cpu")
        self.scaler = None          # bf16 MFMA compute with fp32 master weights: no loss scaling (the reference: fp16 + GradScaler)
        self.alpha = params.get('alpha', 1)
        self.beta = params.get('beta', 0.1)
        self.fds = params.get('fds', False)
        self.distributed = bool(params.get('distributed', False))
        # input pipeline (not reference parameters, defaults keep the reference's in-process loader):
        #   num_workers > 0 -> collate runs in DataLoader worker processes (HostCollate: no model copy in the workers) into
        #   pinned memory; narrow_inputs -> src_edge_type crosses PCIe as int16 and the never-read src_coord stays on the host
        self.num_workers = int(params.get('num_workers', 0))
        self.narrow_inputs = bool(params.get('narrow_inputs', True))
        self.rank = torch.distributed.get_rank() if (self.distributed and torch.distributed.is_initialized()) else 0

    # -------------------------------------------------------------- batches
    def decorate_batch(self, batch, feature_name=None):
        return self.decorate_torch_batch(batch)

    def decorate_torch_batch(self, batch):
        """Host -> device move of a collated batch and the target dtype rule (tasks/trainer.py:101-120)."""
        net_input, net_target = batch
        if isinstance(net_input, dict):
            if self.narrow_inputs:
                net_input = device_payload(net_input, self._n_edge_types, self._pad_idx)
            net_input = to_device(net_input, self.device)
            if self.distributed:
                from ..parallel import pad_to_global_lengths
                net_input = pad_to_global_lengths(net_input)        # the unmasked InfoNCE mean needs one padded length on all ranks
        else:
            net_input = {'net_input': net_input.to(self.device)}
        net_target = net_target.to(self.device, non_blocking=True)
        if self.task == 'repr':
            net_target = None
        elif self.task in ['classification', 'multiclass', 'multilabel_classification']:
            net_target = net_target.long()
        else:
            net_target = net_target.float()
        return net_input, net_target

    _n_edge_types = None
    _pad_idx = 0

    def device_batches(self, dataloader, feature_name=None):
        """(net_input, net_target) on the device for every batch of ``dataloader``, the host-to-device copies of batch i + 1 in flight
        on the copy stream while step i runs (data.DevicePrefetcher) -- the reference moves each batch with synchronous ``.cuda()``
        calls inside the step loop (tasks/trainer.py:181-183).  What decorate_torch_batch does to a batch happens here too, in the
        same order: narrowing, the move, the global padded length under data parallelism, the target dtype rule.  A loader whose
        batches are not the (dict, target) pairs of the model's collate goes through decorate_batch unchanged."""
        import itertools
        it = iter(dataloader)
        try:
            first = next(it)
        except StopIteration:
            return
        batches = itertools.chain([first], it)
        if self.device.type != "cuda" or not isinstance(first[0], dict):
            for batch in batches:
                yield self.decorate_batch(batch, feature_name)
            return
        from ..data import DevicePrefetcher
        for net_input, net_target in DevicePrefetcher(batches, self.device, narrow=self.narrow_inputs, n_edge_types=self._n_edge_types,
                                                      pad_idx=self._pad_idx):
            if self.distributed:
                from ..parallel import pad_to_global_lengths
                net_input = pad_to_global_lengths(net_input)        # the unmasked InfoNCE mean needs one padded length on all ranks
            if self.task == 'repr':
                net_target = None
            elif self.task in ['classification', 'multiclass', 'multilabel_classification']:
                net_target = net_target.long()
            else:
                net_target = net_target.float()
            yield net_input, net_target

    def _collate_for(self, model):
        """The model's own ``batch_collate_fn`` in-process; with worker processes, the same collate as a small picklable
        object (the workers get the pad index and the tokenizer, not the model)."""
        d = getattr(model, 'dictionary', None)
        self._n_edge_types = len(d) * len(d) if d is not None else None
        self._pad_idx = getattr(model, 'padding_idx', 0)
        from ..models.mm_model import MM_Model
        if self.num_workers > 0 and type(model).batch_collate_fn is MM_Model.batch_collate_fn:
            return HostCollate.of(model, narrow=self.narrow_inputs)
        return model.batch_collate_fn

    def _loader_kwargs(self):
        if self.num_workers <= 0:
            return {}
        # (not persistent: a persistent iterator skips the per-epoch base-seed draw, and the shuffle stream would then differ
        # from the in-process loader's -- the epoch order is part of what the G10 fixtures pin)
        # spawned workers, not forked ones: a child forked from a process that has initialised HIP inherits its driver state, and on this
        # stack the parent's GPU work then crawls (measured with bench.py's pipeline workload, 4 forked loader workers: 160-220 ms per step --
        # the host blocked for 0.1-0.35 s at a time in event waits and inside the backward -- against 42 ms with spawned ones and 31 ms of
        # kernels).  Costs a process start-up per loader; the dataset and collate.HostCollate are picklable by construction.
        kw = dict(num_workers=self.num_workers, pin_memory=self.device.type == "cuda")
        if self.device.type == "cuda":
            kw["multiprocessing_context"] = "spawn"
        return kw

    def _require_device(self):
        if self.device.type != "cuda":
            raise RuntimeError("mmdti_hip has no CPU path: construct the Trainer with use_cuda=True on a machine with an MI355X "
                               "(the reference's use_cuda=False plumbing run is the reference's own PyTorch-CPU code)")

    # -------------------------------------------------------------- fit
    def _ddp(self):
        """more than one rank behind this trainer"""
        return self.distributed and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1

    def train_loader(self, model, train_dataset, feature_name=None):
        """-> (loader, sampler | None).  Data parallel: each rank draws a disjoint, equally sized shard of every epoch's permutation
        (``DistributedSampler(drop_last=True)``; batch_size is per rank), so every rank takes the same number of steps -- a rank
        with one batch more would wait forever in that step's collectives.  Device-independent (covered by the 2-rank gloo tests)."""
        if self._ddp():
            from torch.utils.data.distributed import DistributedSampler
            sampler = DistributedSampler(train_dataset, shuffle=True, seed=self.seed, drop_last=True)
            return TorchDataLoader(dataset=train_dataset, batch_size=self.batch_size, sampler=sampler, collate_fn=self._collate_for(model),
                                   drop_last=True, **self._loader_kwargs()), sampler
        return NNDataLoader(feature_name=feature_name, dataset=train_dataset, batch_size=self.batch_size, shuffle=True,
                            collate_fn=self._collate_for(model), drop_last=True, **self._loader_kwargs()), None

    def fit_predict(self, model, train_dataset, valid_dataset, loss_func, activation_fn, dump_dir, fold, target_scaler, feature_name=None,
                    return_infonce_loss=False, return_ct_loss=False, use_weight=False):
        self._require_device()
        model = model.to(self.device)
        train_dataloader, sampler = self.train_loader(model, train_dataset, feature_name)
        min_val_loss, max_score, wait = float("inf"), float("-inf"), 0
        num_training_steps = len(train_dataloader) * self.max_epochs
        engine = FineTuner(model, self.task, learning_rate=self.learning_rate, adam_eps=1e-6, warmup_ratio=0.0,
                           total_steps=num_training_steps, alpha=self.alpha, beta=self.beta,
                           max_norm=self.max_norm if self.amp else None, distributed=self.distributed)
        engine.warmup = int(num_training_steps * self.warmup_ratio)
        self._engine = engine
        task_loss = None if _is_builtin_loss(loss_func, self.task) else loss_func
        self.history = []
        epoch = 0
        for epoch in range(self.max_epochs):
            model = model.train()
            if sampler is not None:
                sampler.set_epoch(epoch)
            start_time = time.time()
            logged = []
            for net_input, net_target in self.device_batches(train_dataloader, feature_name):
                out = engine.step(net_input, net_target, epoch=epoch, use_weight=use_weight, return_infonce_loss=return_infonce_loss,
                                  return_ct_loss=return_ct_loss, loss_func=task_loss)
                logged.append(torch.stack([out.loss, out.task_loss,
                                           out.infonce_loss if out.infonce_loss is not None else out.loss.new_zeros(()),
                                           out.ct_loss if out.ct_loss is not None else out.loss.new_zeros(())]))
            steps = torch.stack(logged).cpu().numpy() if logged else np.zeros((0, 4))        # the epoch's ONE device->host sync
            if self.fds and epoch >= model.fds_cfg.start_update:
                engine.fds_epoch_pass((self.decorate_batch(b, feature_name) for b in train_dataloader), epoch)
            y_preds, val_loss, metric_score = self.predict(model, valid_dataset, loss_func, activation_fn, dump_dir, fold, target_scaler,
                                                           epoch, load_model=False, feature_name=feature_name, return_infonce_loss=False)
            total_val_loss = float(np.mean(val_loss))
            _metric, _score = next(iter(metric_score.items()))
            self.history.append(dict(epoch=epoch, steps=steps, val_loss=total_val_loss, metric=_metric, score=_score))
            if steps.size:
                logger.info('Epoch [{}/{}] train_loss: {:.4f}, train_m_loss: {:.4f}, train_infonce_loss: {:.4f}, train_ct_loss: {:.4f}, '
                            'val_loss: {:.4f}, val_{}: {:.4f}, {:.1f}s'.format(epoch + 1, self.max_epochs, *steps.mean(0), total_val_loss,
                                                                                _metric, _score, time.time() - start_time))
            is_early_stop, min_val_loss, wait, max_score = self._early_stop_choice(
                wait, total_val_loss, min_val_loss, metric_score, max_score, model, dump_dir, fold, self.patience, epoch)
            if is_early_stop:
                break
        self._checkpoint_barrier()          # rank 0's last write is complete before any rank reads the file
        y_preds, _, _ = self.predict(model, valid_dataset, loss_func, activation_fn, dump_dir, fold, target_scaler, epoch, load_model=True,
                                     feature_name=feature_name)
        return y_preds

    # -------------------------------------------------------------- early stopping / checkpoint (:330-385, utils/metrics.py:220-258)
    def _save(self, model, dump_dir, fold):
        """rank 0 only; written to a temporary name and renamed, so a reader never sees a half-written archive (torch.save is
        not atomic)"""
        if self.rank != 0:
            return
        os.makedirs(dump_dir, exist_ok=True)
        path = os.path.join(dump_dir, f'model_{fold}.pth')
        tmp = path + f'.tmp{os.getpid()}'
        torch.save({'model_state_dict': model.state_dict()}, tmp)
        os.replace(tmp, path)

    def _checkpoint_barrier(self):
        if self._ddp():
            from ..parallel import host_group
            torch.distributed.barrier(group=host_group())

    def _agree(self, flag: bool) -> bool:
        """Data parallel: every rank follows RANK 0's reading of a per-epoch decision.  Each rank validates on its own device and the
        loss kernels add with atomics, so two ranks can differ in the last bit of a validation scalar; at a near-tie one rank
        would leave the epoch loop while the others enter the next step's collectives and wait forever."""
        if not self._ddp():
            return flag
        from ..parallel import host_broadcast_ints
        return bool(host_broadcast_ints([int(flag)], src=0)[0])

    def _early_stop_choice(self, wait, loss, min_loss, metric_score, max_score, model, dump_dir, fold, patience, epoch):
        by_loss = not isinstance(self.metrics_str, str) or self.metrics_str in ['loss', 'none', '']
        if by_loss:
            value, best, increase = loss, min_loss, False
        else:
            name, value = next(iter(metric_score.items()))
            increase = _is_increase(self.metrics, self.task, name)
            best = max_score if increase else min_loss
        improved = self._agree(value >= best if increase else value <= best)
        stop = False
        if improved:
            best, wait = value, 0
            self._save(model, dump_dir, fold)
        else:
            wait += 1
            if wait == patience:
                logger.warning(f'Early stopping at epoch: {epoch + 1}')
                stop = True
        if increase:
            return stop, min_loss, wait, best
        return stop, best, wait, max_score

    # -------------------------------------------------------------- predict
    def predict(self, model, dataset, loss_func, activation_fn, dump_dir, fold, target_scaler=None, epoch=1, load_model=False,
                feature_name=None, return_infonce_loss=False, return_ct_loss=False, return_feature=False):
        self._require_device()
        model = model.to(self.device)
        if load_model == True:      # noqa: E712  (the reference's spelling; callers pass bools)
            sd = torch.load(os.path.join(dump_dir, f'model_{fold}.pth'), map_location=self.device, weights_only=True)["model_state_dict"]
            model.load_state_dict(sd)           # (an arena-bound model re-casts its bf16 weight shadow on the next GEMM: runtime._fresh)
        dataloader = NNDataLoader(feature_name=feature_name, dataset=dataset, batch_size=self.batch_size, shuffle=False,
                                  collate_fn=self._collate_for(model), **self._loader_kwargs())
        model = model.eval()
        val_loss, y_preds, y_truths = [], [], []
        builtin = _is_builtin_loss(loss_func, self.task)
        with torch.no_grad():
            for net_input, net_target in self.device_batches(dataloader, feature_name):
                outputs = model(**net_input)       # both auxiliary losses are force-disabled in the reference's predict (:427-428)
                if not load_model:
                    tl = _builtin_loss(self.task)(outputs, net_target) if builtin else loss_func(outputs, net_target)
                    val_loss.append(self.alpha * tl)
                y_preds.append(activation_fn(outputs))
                y_truths.append(net_target)
        y_preds = torch.cat(y_preds).float().cpu().numpy()
        y_truths = torch.cat(y_truths).cpu().numpy()
        val_loss = [float(v) for v in torch.stack(val_loss).cpu()] if val_loss else []
        label_cnt = getattr(model, "output_dim", None)
        metric_score = None
        if not load_model:
            if self.alpha != 0:
                if target_scaler is not None:
                    metric_score = self.metrics.cal_metric(target_scaler.inverse_transform(y_truths), target_scaler.inverse_transform(y_preds),
                                                           label_cnt=label_cnt)
                else:
                    metric_score = self.metrics.cal_metric(y_truths, y_preds, label_cnt=label_cnt)
            else:
                metric_score = {"ct_loss": float(np.mean(val_loss))}
        elif self.alpha == 0:
            metric_score = {"ct_loss": float(np.mean(val_loss)) if val_loss else float("nan")}
        return y_preds, val_loss, metric_score

    def set_seed(self, seed):
        torch.manual_seed(seed)
        if torch.cuda.is_available():
            torch.cuda.manual_seed_all(seed)
        np.random.seed(seed)


def _builtin_loss(task):
    from ..functional import CELossFn, MSELossFn, BCELogitsLossFn
    if task == "regression":
        return lambda o, t: MSELossFn.apply(o, t.float())
    if task == "multilabel_classification":
        return lambda o, t: BCELogitsLossFn.apply(o, t)
    return lambda o, t: CELossFn.apply(o, t)


def _is_builtin_loss(loss_func, task):
    """True when ``loss_func`` is the reference's task loss for this task (models/nnmodel.py:24-34: ``nn.MSELoss()`` /
    ``myCrossEntropyLoss``) -- those run in the mse / cross-entropy kernels; any other callable is applied as given."""
    if loss_func is None:
        return task in ("regression", "classification", "multiclass", "multilabel_classification")
    if task == "regression":
        return isinstance(loss_func, torch.nn.MSELoss) and loss_func.reduction == "mean"
    if task in ("classification", "multiclass"):
        if getattr(loss_func, "__name__", "") == "myCrossEntropyLoss":
            return True
        # a configured nn.CrossEntropyLoss (class weights, label smoothing, another ignore_index / reduction) is NOT the plain kernel
        return (isinstance(loss_func, torch.nn.CrossEntropyLoss) and loss_func.weight is None and loss_func.label_smoothing == 0.0
                and loss_func.ignore_index == -100 and loss_func.reduction == "mean")
    if task == "multilabel_classification":      # the table's 'bce' entry; 'focal' / 'ghm' are applied as given
        return (isinstance(loss_func, torch.nn.BCEWithLogitsLoss) and loss_func.weight is None and loss_func.pos_weight is None
                and loss_func.reduction == "mean")
    return False


def NNDataLoader(feature_name=None, dataset=None, batch_size=None, shuffle=False, collate_fn=None, drop_last=False, **loader_kwargs):
    """tasks/trainer.py:540-556 (``loader_kwargs``: num_workers / pin_memory / persistent_workers, not in the reference)."""
    return TorchDataLoader(dataset=dataset, batch_size=batch_size, shuffle=shuffle, collate_fn=collate_fn, drop_last=drop_last, **loader_kwargs)
