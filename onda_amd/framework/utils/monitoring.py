"""Drop-in for the hot-path part of ``framework/utils/monitoring.py``: ``Monitor`` (:7-96), the windowed statistics
behind the static / dynamic switch.

Same call surface and values as the reference (fixture G5): ``add({key: value})``, ``avg`` = median of the last
``limit`` samples, ``exp`` = exponential moving average started at the first sample, ``dev_avg`` = weighted level of
the window without its oldest sample minus that of the window without its newest one (zero until the window is
full), ``eval()`` / ``train()`` freeze and unfreeze it.  Built differently: every key owns a fixed ring of float64
slots (no list growth, no ``pop(0)``), samples are plain floats (a device scalar given to ``add`` is read back once,
there), and several scalars that live on the device can be added with ONE transfer (``add_device``) that does not
stop the host: the copy to pinned memory is enqueued and the samples enter their rings when the monitor is next
looked at or written to (by then, normally, the copy has long finished).
"""
import os

import numpy as np
import torch

_SYNC_TRANSFERS = os.environ.get("ONDA_MONITOR_SYNC", "0") == "1"  # measurement knob: wait for every transfer at once


class _Series:
    """Ring buffer of one monitored quantity."""
    __slots__ = ("ring", "count", "head", "ema")

    def __init__(self, capacity, first):
        self.ring = np.empty(capacity, dtype=np.float64)
        self.ring[0] = first
        self.count, self.head, self.ema = 1, 1 % capacity, first

    def push(self, value, smoothing):
        self.ring[self.head] = value
        self.head = (self.head + 1) % self.ring.size
        self.count = min(self.count + 1, self.ring.size)
        self.ema = (1 - smoothing) * self.ema + smoothing * value

    def ordered(self):
        """Oldest -> newest."""
        if self.count < self.ring.size:
            return self.ring[: self.count]
        return np.concatenate([self.ring[self.head:], self.ring[: self.head]])


def _level_function(kind, span):
    if kind == "median":
        return lambda w: float(np.median(w))
    if kind == "mean":
        return lambda w: float(np.mean(w))
    if kind == "hamming":
        taps = np.hamming(span)
        total = np.sum(taps)
        return lambda w: np.sum(taps * w) / total
    raise ValueError(f"unknown DEV_MONITOR_FUNC {kind!r}")


class Monitor(object):
    def __init__(self, limit=None, exp_const=0.01, dev_func="hamming"):
        self.limit = limit
        self.exp_const = exp_const
        self.freeze = False
        self._series = {}
        self._level = _level_function(dev_func, limit - 1)
        self._unbounded = 1 << 16  # limit None: the reference keeps every sample
        self._pending = None       # (keys, pinned host tensor, event) of an add_device whose copy may still be in flight
        self._pinned = []          # two pinned staging buffers, used alternately

    # ---- mode ------------------------------------------------------------------------------------------------------
    def eval(self):
        self.freeze = True

    def train(self):
        self.freeze = False

    def reset(self):
        self._pending = None
        self._series = {}

    def _flush(self, item=None):
        """Samples of the last add_device enter their rings (waits for its copy if that has not finished).  A query
        for one series that the pending transfer does not feed leaves it alone."""
        if self._pending is None or (item is not None and item not in self._pending[0]):
            return
        keys, host, event = self._pending
        self._pending = None
        event.synchronize()
        for key, value in zip(keys, host[: len(keys)].tolist()):
            self._push(key, value)

    # ---- samples ---------------------------------------------------------------------------------------------------
    def _push(self, key, value, reset=False):
        s = self._series.get(key)
        if s is None or reset:
            self._series[key] = _Series(self.limit or self._unbounded, value)
        else:
            s.push(value, self.exp_const)

    def add(self, values, reset=False):
        if self.freeze:
            return 0
        self._flush()
        for key, value in values.items():
            self._push(key, value.item() if isinstance(value, torch.Tensor) else value, reset)

    def add_device(self, keys, packed):
        """keys[i] <- packed[i] for a 1-D device tensor `packed`: one read-back for all of them."""
        if self.freeze:
            return 0
        self._flush()
        if not packed.is_cuda:
            for key, value in zip(keys, packed.tolist()):
                self._push(key, value)
            return
        n = packed.numel()
        if len(self._pinned) < 2 or self._pinned[0].numel() < n:
            self._pinned = [torch.empty(max(n, 16), dtype=torch.float32, pin_memory=True) for _ in range(2)]
        host = self._pinned.pop(0)
        self._pinned.append(host)
        host[:n].copy_(packed.detach().reshape(-1).float(), non_blocking=True)
        event = torch.cuda.Event()
        event.record()
        self._pending = (list(keys), host, event)
        if _SYNC_TRANSFERS:
            self._flush()

    # ---- statistics ------------------------------------------------------------------------------------------------
    def avg(self, item=None):
        self._flush(item)
        if item is not None:
            s = self._series.get(item)
            return float(np.median(s.ordered())) if s is not None else 1
        return {key: float(np.median(s.ordered())) for key, s in self._series.items()}

    def exp(self, item=None):
        self._flush(item)
        if item is not None:
            s = self._series.get(item)
            return s.ema if s is not None else 1
        return {key: s.ema for key, s in self._series.items()}

    def _trend(self, s):
        if s is None or self.limit is None or s.count < self.limit:
            return 0
        w = s.ordered()
        return self._level(w[1:]) - self._level(w[:-1])

    def dev_avg(self, item=None):
        self._flush(item)
        if item is not None:
            return self._trend(self._series.get(item))
        return {key: self._trend(s) for key, s in self._series.items()}

    # the reference's attribute names, for code that peeks at them
    @property
    def current_dict(self):
        self._flush()
        return {key: list(s.ordered()) for key, s in self._series.items()}

    @property
    def exp_dict(self):
        self._flush()
        return {key: s.ema for key, s in self._series.items()}
