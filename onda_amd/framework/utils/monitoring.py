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

_SYNC_TRANSFERS = False  # (tests: wait for every transfer at once)


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
        self.dev_func = dev_func
        self.freeze = False
        self._series = {}
        self._level = _level_function(dev_func, limit - 1)
        self._unbounded = 1 << 16  # limit None: the reference keeps every sample
        self._pending = []         # FIFO of (keys, pinned host tensor, event): add_device transfers not yet in the rings
        self._pinned = []          # free pinned staging buffers
        self.gated = set()         # keys whose device samples may be NaN = "no sample this step" (DeviceSwitch)

    # ---- mode ------------------------------------------------------------------------------------------------------
    def eval(self):
        self.freeze = True

    def train(self):
        self.freeze = False

    def reset(self):
        self._pending = []
        self._series = {}

    def _flush(self, item=None, block=True):
        """Samples of finished add_device transfers enter their rings, oldest transfer first.  block=True waits for every
        transfer (a statistic is about to be read); a query for ONE series that no pending transfer feeds waits for
        nothing.  block=False takes what has arrived and stops at the first transfer still in flight (order is kept)."""
        if not self._pending or (item is not None and not any(item in p[0] for p in self._pending)):
            return
        while self._pending:
            keys, host, event = self._pending[0]
            if not block and not event.query():
                return
            event.synchronize()
            self._pending.pop(0)
            for key, value in zip(keys, host[: len(keys)].tolist()):
                if value != value and key in self.gated:
                    continue  # NaN on a gated key: the quantity did not exist this step
                self._push(key, value)
            self._pinned.append(host)

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
        self._flush()  # (host samples enter behind everything added before them)
        for key, value in values.items():
            self._push(key, value.item() if isinstance(value, torch.Tensor) else value, reset)

    def add_device(self, keys, packed):
        """keys[i] <- packed[i] for a 1-D device tensor `packed`: one read-back for all of them, never waited for here."""
        if self.freeze:
            return 0
        self._flush(block=False)
        if not packed.is_cuda:
            self._flush()
            for key, value in zip(keys, packed.tolist()):
                if not (value != value and key in self.gated):
                    self._push(key, value)
            return
        n = packed.numel()
        host = next((h for h in self._pinned if h.numel() >= n), None)
        if host is not None:
            self._pinned.remove(host)
        else:
            host = torch.empty(max(n, 16), dtype=torch.float32, pin_memory=True)
        host[:n].copy_(packed.detach().reshape(-1).float(), non_blocking=True)
        event = torch.cuda.Event()
        event.record()
        self._pending.append((list(keys), host, event))
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


class DeviceSwitch:
    """``Monitor`` for ONE series plus ``model_select`` (prototypes_hybrid_switch.py:22-34), kept and advanced on the
    GPU: ``step(sample)`` = add the sample, take median / exponential average / trend, run the two-state machine -- one
    small launch, no read-back.  ``flag`` (device int32[1]) is the machine's state: 0 static, 1 dynamic."""

    def __init__(self, device, limit, exp_const=0.01, dev_func="hamming", gray_area=(0.84, 0.88), dev_threshold=0.0002,
                 start=0, use_exp=False):
        from onda_amd import _lib
        from onda_amd._lib import OndaSwitchCfg, query
        if dev_func not in ("hamming", "mean", "median"):
            raise ValueError(f"unknown DEV_MONITOR_FUNC {dev_func!r}")
        span = max(limit - 1, 1)
        taps = np.hamming(span) if dev_func == "hamming" else np.ones(span)
        self.cfg = OndaSwitchCfg(int(limit), 1 if dev_func == "median" else 0, int(bool(use_exp)), 0, float(exp_const),
                                 float(1 - exp_const), float(np.sum(taps)), float(gray_area[0]), float(gray_area[1]),
                                 float(dev_threshold))
        self.taps = torch.from_numpy(taps.astype(np.float64)).to(device)
        self.state = torch.zeros(query("onda_switch_state_doubles", int(limit)), dtype=torch.float64, device=device)
        self.istate = torch.zeros(8, dtype=torch.int32, device=device)
        self.istate[2:4] = int(start)
        self.flag = torch.full((1,), int(start), dtype=torch.int32, device=device)
        self._lib = _lib

    def step(self, sample):
        """`sample`: device scalar, float32 (or float64: the scripted sequences of the tests)."""
        from ctypes import byref
        sample = sample.detach().reshape(1)
        if sample.dtype not in (torch.float32, torch.float64):
            sample = sample.float()
        self._lib.call("onda_switch_step", self.state.data_ptr(), self.istate.data_ptr(), sample.data_ptr(),
                       int(sample.dtype == torch.float64), self.taps.data_ptr(), byref(self.cfg), self.flag.data_ptr(),
                       torch.cuda.current_stream().cuda_stream)
        return self.flag

    # blocking reads: for tests and for code that looks at the switch between steps (never on the step's own path)
    def read(self):
        st, ist = self.state[:4].tolist(), self.istate[:5].tolist()
        return {"exp": st[0], "avg": st[1], "dev": st[2], "confidence": st[3], "count": ist[0], "current": ist[2],
                "current_dev": ist[3], "steps": ist[4]}
