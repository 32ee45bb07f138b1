"""ORACLE (test infrastructure): host-side statistics that drive the static/dynamic switch.

Parity status: PINNED by fixture G5 (scripted scalar sequences run through the
reference's ``Monitor`` and ``model_select``).

* WindowStats  framework/utils/monitoring.py:7-96  (median / exp-MA / Hamming derivative)
* SwitchState  framework/domain_adaptation/methods/prototypes_hybrid_switch.py:5-34
"""
from statistics import median

import numpy as np


class WindowStats:
    def __init__(self, limit=200, exp_const=0.01, dev_func="hamming"):
        self.limit, self.exp_const, self.freeze = limit, exp_const, False
        self.window, self.ema = {}, {}
        self.kernel = np.hamming(limit - 1)
        if dev_func == "median":
            self.level = median
        elif dev_func == "mean":
            self.level = lambda v: float(np.mean(np.asarray(v)))
        else:
            self.level = lambda v: np.sum(self.kernel * np.asarray(v)) / np.sum(self.kernel)

    def add(self, values):
        if self.freeze:
            return
        for k, v in values.items():
            if k not in self.window:
                self.window[k], self.ema[k] = [v], v
                continue
            self.window[k].append(v)
            if len(self.window[k]) > self.limit:
                self.window[k].pop(0)
            self.ema[k] = (1 - self.exp_const) * self.ema[k] + self.exp_const * v

    def avg(self, key=None):
        if key is None:
            return {k: median(v) for k, v in self.window.items()}
        return median(self.window[key]) if key in self.window else 1

    def exp(self, key=None):
        if key is None:
            return self.ema
        return self.ema.get(key, 1)

    def dev_avg(self, key):
        w = self.window.get(key)
        if w is None or len(w) < self.limit:
            return 0
        return self.level(w[1:]) - self.level(w[:-1])


class SwitchState:
    STATIC, DYNAMIC = 0, 1

    def __init__(self, gray_area=(0.84, 0.88), dev_threshold=2e-4):
        self.current = self.trend = self.STATIC
        self.gray_area, self.dev_threshold, self.freeze = gray_area, dev_threshold, False

    def evaluate(self, confidence, dev_value):
        if self.freeze:
            return
        if dev_value > self.dev_threshold:
            self.trend = self.STATIC
        elif dev_value < -self.dev_threshold:
            self.trend = self.DYNAMIC
        if confidence < self.gray_area[0]:
            self.current = self.DYNAMIC
        elif confidence > self.gray_area[1]:
            self.current = self.STATIC
        else:
            self.current = self.trend
