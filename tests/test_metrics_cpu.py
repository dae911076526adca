"""EpochMetric (mmdti_hip/tasks/trainer.py): the metrics the Trainer drop-in computes itself agree with the functions the
reference's utils/metrics.py:72-112 table names (sklearn / scipy), and the first default metric per task is the reference's
(utils/metrics.py:114-120)."""
import numpy as np
import pytest

from mmdti_hip.tasks import trainer as t


def test_metric_functions_match_sklearn_and_scipy():
    M = pytest.importorskip("sklearn.metrics")
    st = pytest.importorskip("scipy.stats")
    rng = np.random.default_rng(0)
    y = (rng.random(300) < 0.3).astype(int)
    p = np.round(rng.random(300), 2)                                   # ties included
    q = (p > 0.5).astype(int)
    for name, ref in (("auc", M.roc_auc_score(y, p)), ("auprc", M.average_precision_score(y, p)),
                      ("f1_score", M.f1_score(y, q)), ("mcc", M.matthews_corrcoef(y, q)), ("acc", M.accuracy_score(y, q))):
        assert abs(t._METRIC_TABLE[name][0](y, p) - ref) < 1e-12, name
    # log loss: the reference hands sklearn float32 predictions (utils/metrics.py:168), whose eps clips saturated probabilities
    assert abs(t._METRIC_TABLE["log_loss"][0](y, p) - M.log_loss(y, p.astype(np.float32))) < 1e-6
    ys, ps = np.array([0, 1, 1, 0, 1]), np.array([1.0, 1.0, 0.7, 0.2, 0.0], dtype=np.float32)       # saturated softmax outputs
    assert abs(t._METRIC_TABLE["log_loss"][0](ys, ps) - M.log_loss(ys, ps)) < 1e-5 and abs(M.log_loss(ys, ps) - 6.49) < 0.01
    a = rng.normal(size=200)
    b = np.round(a + rng.normal(size=200), 1)
    for name, ref in (("pearsonr", st.pearsonr(a, b)[0]), ("spearmanr", st.spearmanr(a, b)[0]), ("mse", M.mean_squared_error(a, b)),
                      ("mae", M.mean_absolute_error(a, b)), ("r2", M.r2_score(a, b))):
        assert abs(t._METRIC_TABLE[name][0](a, b) - ref) < 1e-10, name


def test_default_metric_and_direction_follow_the_reference():
    # utils/metrics.py:114-120: the first default metric per task; :72-112: the direction of each
    for task, first, inc in (("regression", "mse", False), ("classification", "log_loss", False), ("multilabel_classification", "log_loss", False)):
        for spelled in ("none", "", None):
            m = t.EpochMetric(task, spelled)
            assert m.names == [first] and m.is_increase(first) is inc
    m = t.EpochMetric("classification", "auprc,mcc")
    assert m.names == ["auprc", "mcc"] and m.is_increase("auprc")
    y = np.array([[0], [1], [1], [0]]); p = np.array([[0.1], [0.8], [0.6], [0.4]])
    out = m.cal_metric(y, p)
    assert list(out) == ["auprc", "mcc"] and out["auprc"] == 1.0 and out["mcc"] == 1.0
    with pytest.raises(ValueError):
        t.EpochMetric("classification", "cohen_kappa")
