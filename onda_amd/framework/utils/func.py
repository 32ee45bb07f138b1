"""Drop-in for the hot-path helpers of ``framework/utils/func.py``: ``loss_calc`` (:35-42),
``lr_poly`` (:45-47) and the evaluation helpers ``fast_hist`` / ``per_class_iu`` (:77-85)."""
import numpy as np

from .loss import cross_entropy_2d


def loss_calc(pred, label, device, soft=False):
    return cross_entropy_2d(pred, label.long().to(device), soft)


def lr_poly(base_lr, iter, max_iter, power):
    return base_lr * ((1 - float(iter) / max_iter) ** power)


def fast_hist(a, b, n):
    k = (a >= 0) & (a < n)
    return np.bincount(n * a[k].astype(int) + b[k], minlength=n ** 2).reshape(n, n)


def per_class_iu(hist):
    return np.diag(hist) / (hist.sum(1) + hist.sum(0) - np.diag(hist) + np.finfo(float).eps)
