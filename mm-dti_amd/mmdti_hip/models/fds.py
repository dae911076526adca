"""Drop-in for the reference's ``models/fds.py`` (feature distribution smoothing), MI355X-native.

Same class name, ctor arguments, the 8 registered buffers (``state_dict`` round-trips with the reference's checkpoints)
and methods ``smooth / update_last_epoch_stats / update_running_stats / reset`` as /root/reference/models/fds.py:31-190,
plus module-level ``anomaly_clean_regression`` (:18-29).  Binning (fp32 floor division, :125/:164), bucket membership
(first bucket takes ``<=``, last takes ``>=``, only for buckets present in the batch), per-bucket mean / unbiased
variance, momentum EMA, reflect-padded kernel smoothing and ``calibrate_mean_var`` (utils/util.py:159-169) all run in
gfx950 kernels; the reference does the binning in a per-sample Python list comprehension and one masked index_put per
bucket.

Differences by design: no hard-coded ``'cuda'`` (:84) -- buffers follow the module's device; ``raw_data`` may be a CSV
path (as in the reference) or an array of the training targets.
"""
import copy

import numpy as np
import torch
import torch.nn as nn

from .. import ops
from ..functional import FDSSmoothFn


def anomaly_clean_regression(data):
    """3-sigma cleaning (models/fds.py:18-29)."""
    _mean, _std = data.mean(), data.std()
    data = data[(data > _mean - 3 * _std) & (data < _mean + 3 * _std)]
    return data


def _kernel_window(kernel, ks, sigma):
    """The ks-tap smoothing window over label buckets (values of FDS._get_kernel_window, fds.py:69-84; the four windows
    the reference can produce are pinned by tests/golden/g4_fds_*.npz).  All three shapes are unit-sum:
      gaussian : scipy's truncated-at-4-sigma, reflect-extended Gaussian filter applied to a centred unit impulse of length
                 ks (for ks < 8*sigma the reflected tails fold back into the window -- that folding is part of the values);
      triang   : scipy's triangular window;
      laplace  : exp(-|k|/sigma) on k = -half..half."""
    from scipy.ndimage import gaussian_filter1d
    from scipy.signal.windows import triang
    half = (ks - 1) // 2
    if kernel == 'gaussian':
        impulse = np.zeros(2 * half + 1, dtype=np.float32)
        impulse[half] = 1.0
        taps = gaussian_filter1d(impulse, sigma=sigma)
    elif kernel == 'triang':
        taps = triang(ks)
    elif kernel == 'laplace':
        taps = np.exp(-np.abs(np.arange(-half, half + 1)) / sigma) / (2.0 * sigma)
    else:
        raise AssertionError(f"FDS kernel must be gaussian, triang or laplace, got {kernel!r}")
    return torch.tensor(np.asarray(taps / sum(taps)), dtype=torch.float32)


class FDS(nn.Module):

    def __init__(self, feature_dim, raw_data, col_data, using_scale, bucket_num=100, bucket_start=0, start_update=0, start_smooth=1,
                 kernel='gaussian', ks=5, sigma=2, momentum=0.9, device=None):
        super(FDS, self).__init__()
        self.feature_dim = feature_dim
        self.bucket_num = bucket_num
        self.bucket_start = bucket_start
        self.half_ks = (ks - 1) // 2
        self.momentum = momentum
        self.start_update = start_update
        self.start_smooth = start_smooth
        if momentum is None:
            raise NotImplementedError("momentum=None (cumulative average) is not on the MM-DTI path (fds_config: 0.9)")
        if isinstance(raw_data, str):
            import pandas as pd
            self.raw_data = pd.read_csv(raw_data).loc[:, col_data].values
        else:
            self.raw_data = np.asarray(raw_data)
        regression_value = copy.deepcopy(self.raw_data).astype(np.float64)
        if using_scale:
            regression_value = (regression_value - regression_value.mean()) / regression_value.std()    # StandardScaler
            regression_value = anomaly_clean_regression(regression_value)
        value_range = np.max(regression_value) - np.min(regression_value)
        self.min_value = np.min(regression_value)
        self.bin_width = value_range / bucket_num
        self.register_buffer('kernel_window', _kernel_window(kernel, ks, sigma), persistent=False)
        self.register_buffer('epoch', torch.zeros(1).fill_(start_update))
        self.register_buffer('running_mean', torch.zeros(bucket_num - bucket_start, feature_dim))
        self.register_buffer('running_var', torch.ones(bucket_num - bucket_start, feature_dim))
        self.register_buffer('running_mean_last_epoch', torch.zeros(bucket_num - bucket_start, feature_dim))
        self.register_buffer('running_var_last_epoch', torch.ones(bucket_num - bucket_start, feature_dim))
        self.register_buffer('smoothed_mean_last_epoch', torch.zeros(bucket_num - bucket_start, feature_dim))
        self.register_buffer('smoothed_var_last_epoch', torch.ones(bucket_num - bucket_start, feature_dim))
        self.register_buffer('num_samples_tracked', torch.zeros(bucket_num - bucket_start))
        if device is not None:
            self.to(device)

    def _bins(self, labels):
        l0 = labels[:, 0] if labels.dim() > 1 else labels
        l0 = l0.detach().to(self.running_mean.device, torch.float32).contiguous()
        return ops.fds_bins(l0, float(self.min_value), float(self.bin_width), self.bucket_start, self.bucket_num)

    def _update_last_epoch_stats(self):
        # the reference aliases running_* into *_last_epoch (:87-88); values are what matters for state_dict parity
        self.running_mean_last_epoch = self.running_mean
        self.running_var_last_epoch = self.running_var
        self.smoothed_mean_last_epoch = ops.fds_smooth_stats(self.running_mean_last_epoch, self.kernel_window)
        self.smoothed_var_last_epoch = ops.fds_smooth_stats(self.running_var_last_epoch, self.kernel_window)

    def reset(self):
        self.running_mean.zero_()
        self.running_var.fill_(1)
        self.running_mean_last_epoch.zero_()
        self.running_var_last_epoch.fill_(1)
        self.smoothed_mean_last_epoch.zero_()
        self.smoothed_var_last_epoch.fill_(1)
        self.num_samples_tracked.zero_()

    def update_last_epoch_stats(self, epoch):
        if epoch == float(self.epoch) + 1:
            self.epoch += 1
            self._update_last_epoch_stats()

    def update_running_stats(self, features, labels, epoch):
        if epoch < float(self.epoch):
            return
        assert self.feature_dim == features.size(1), "Input feature dimension is not aligned!"
        assert features.size(0) == labels.size(0), "Dimensions of features and labels are not aligned!"
        bins, flags = self._bins(labels)
        factor = 0.0 if epoch == self.start_update else self.momentum
        feats = features.detach().to(self.running_mean.device, torch.float32).contiguous()
        ops.fds_update_stats(feats, bins, flags, self.bucket_start, self.bucket_num, factor, self.running_mean, self.running_var,
                             self.num_samples_tracked)

    def smooth(self, features, labels, epoch):
        if epoch < self.start_smooth:
            return features
        bins, flags = self._bins(labels)
        y = FDSSmoothFn.apply(features, bins, flags, self.bucket_start, self.bucket_num, self.running_mean_last_epoch,
                              self.running_var_last_epoch, self.smoothed_mean_last_epoch, self.smoothed_var_last_epoch)
        return y
