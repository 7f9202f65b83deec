"""numpy metrics with the semantics of the sklearn functions the reference's `compile` wires in
(deepctr/models/basemodel.py:496-512: log_loss, roc_auc_score, mean_squared_error, accuracy)."""
import numpy as np
import torch


def log_loss(y_true, y_pred):
    """Binary cross-entropy, mean over samples; probabilities clipped to [eps, 1-eps] with
    eps = machine epsilon of y_pred's dtype (sklearn >= 1.5 behaviour; the reference always passes
    float64 predictions, basemodel.py:269,352)."""
    y = np.asarray(y_true, dtype=np.float64).ravel()
    p = np.asarray(y_pred)
    eps = np.finfo(p.dtype if p.dtype.kind == "f" else np.float64).eps
    p = np.clip(p.astype(np.float64).ravel(), eps, 1 - eps)
    return float(-np.mean(y * np.log(p) + (1 - y) * np.log(1 - p)))


def roc_auc_score(y_true, y_score):
    """Area under the ROC curve for binary labels (Mann-Whitney statistic with mid-ranks for ties)."""
    y = np.asarray(y_true).ravel() > 0.5
    s = np.asarray(y_score, dtype=np.float64).ravel()
    n_pos = int(y.sum())
    n_neg = int(y.size - n_pos)
    if n_pos == 0 or n_neg == 0:
        raise ValueError("Only one class present in y_true. ROC AUC score is not defined in that case.")
    order = np.argsort(s, kind="mergesort")
    ss = s[order]
    # mid-rank of each tie group
    boundaries = np.flatnonzero(np.concatenate(([True], ss[1:] != ss[:-1], [True])))
    starts, ends = boundaries[:-1], boundaries[1:]
    mid = 0.5 * (starts + ends - 1) + 1.0
    ranks = np.empty(s.size, dtype=np.float64)
    ranks[order] = np.repeat(mid, ends - starts)
    return float((ranks[y].sum() - n_pos * (n_pos + 1) / 2.0) / (float(n_pos) * n_neg))


def mean_squared_error(y_true, y_pred):
    d = np.asarray(y_true, dtype=np.float64).ravel() - np.asarray(y_pred, dtype=np.float64).ravel()
    return float(np.mean(d * d))


def accuracy_score(y_true, y_pred):
    return float(np.mean(np.asarray(y_true).ravel() == np.asarray(y_pred).ravel()))


# ------------------------------------------------------------------------------------------------------------------
# The same metrics on device tensors, each returning a 0-d float64 tensor without any host synchronisation: `fit`
# logs them per step (basemodel.py:264-269 calls the sklearn functions on host copies of every batch -- a device
# to host copy, a sync and a CPU sort per step) and reads the epoch's values back once.
# ------------------------------------------------------------------------------------------------------------------
def log_loss_device(y_true, y_pred):
    y = y_true.reshape(-1).double()
    eps = float(np.finfo(np.float64).eps)              # the reference hands float64 predictions to sklearn
    p = y_pred.reshape(-1).double().clamp(eps, 1 - eps)
    return -(y * torch.log(p) + (1 - y) * torch.log(1 - p)).mean()


def roc_auc_score_device(y_true, y_score):
    """Mann-Whitney statistic with mid-ranks, as roc_auc_score above: ranks are multiples of 0.5, so their float64 sum
    is exact in any order and the result equals the numpy version bit for bit.  A batch with a single class gives NaN
    (the host version raises there; `fit` raises the same error when it reads the epoch's values)."""
    pos = (y_true.reshape(-1) > 0.5).double()
    s = y_score.reshape(-1).double()
    ss, _ = torch.sort(s)
    lo = torch.searchsorted(ss, s, right=False)
    hi = torch.searchsorted(ss, s, right=True)
    ranks = 0.5 * (lo + hi - 1).double() + 1.0
    n_pos = pos.sum()
    n_neg = pos.numel() - n_pos
    return ((ranks * pos).sum() - n_pos * (n_pos + 1) / 2.0) / (n_pos * n_neg)


def mean_squared_error_device(y_true, y_pred):
    d = y_true.reshape(-1).double() - y_pred.reshape(-1).double()
    return (d * d).mean()


DEVICE = {log_loss: log_loss_device, roc_auc_score: roc_auc_score_device, mean_squared_error: mean_squared_error_device}
