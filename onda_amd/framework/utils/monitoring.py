"""Drop-in for the parts of ``framework/utils/monitoring.py`` on the hot path: ``Monitor``
(:7-96), the windowed statistics that drive the static/dynamic switch.

Same API and values as the reference; the difference is that samples are stored as Python
floats (a device scalar is read back once, when it is added) instead of 0-dim device
tensors whose every comparison inside ``statistics.median`` is a host sync.
"""
from statistics import median

import numpy as np
import torch


def _scalar(v):
    if isinstance(v, torch.Tensor):
        return v.item()
    return v


class Monitor(object):
    def __init__(self, limit=None, exp_const=0.01, dev_func="hamming"):
        self.current_dict = {}
        self.limit = limit
        self.exp_dict = {}
        self.exp_const = exp_const
        self.freeze = False
        self.signal = np.hamming(limit - 1)
        self.signal_sum = np.sum(self.signal)
        if dev_func == "median":
            self.mean_func = median
        elif dev_func == "mean":
            self.mean_func = lambda x: np.mean(np.array(x))
        elif dev_func == "hamming":
            self.mean_func = lambda x: np.sum(self.signal * np.array(x)) / self.signal_sum

    def eval(self):
        self.freeze = True

    def train(self):
        self.freeze = False

    def add(self, values, reset=False):
        if self.freeze:
            return 0
        for key, val in values.items():
            val = _scalar(val)
            window = self.current_dict.get(key)
            if window is None or reset:
                self.current_dict[key] = [val]
                self.exp_dict[key] = val
                continue
            window.append(val)
            if self.limit is not None and len(window) > self.limit:
                window.pop(0)
            self.exp_dict[key] = (1 - self.exp_const) * self.exp_dict[key] + self.exp_const * val

    def _dev_avg(self, item):
        window = self.current_dict.get(item)
        if window is None or len(window) < self.limit:
            return 0
        return self.mean_func(window[1:]) - self.mean_func(window[:-1])

    def dev_avg(self, item=None):
        if item is not None:
            return self._dev_avg(item)
        return {key: self._dev_avg(key) for key in self.current_dict}

    def exp(self, item=None):
        if item is not None:
            return self.exp_dict.get(item, 1)
        return self.exp_dict

    def avg(self, item=None):
        if item is not None:
            return median(self.current_dict[item]) if item in self.current_dict else 1
        return {key: median(vals) for key, vals in self.current_dict.items()}

    def reset(self):
        self.current_dict = {}
