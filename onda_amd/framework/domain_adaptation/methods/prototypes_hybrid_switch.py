"""Drop-in for ``framework/domain_adaptation/methods/prototypes_hybrid_switch.py``: ``model_select`` (:5-34) and
``hybrid_proDA`` (:37-109), the hybrid static/dynamic switch of ``configs/hybrid_switch.yml``.

The method differs from ``online_proDA`` only in how the priors are mixed, so it only supplies ``_prior_plan``: the
step's machinery (pipelined teacher / static / student passes, deferred monitor entries, gradient exchange) is the
base class's.

On the GPU the switch itself lives on the device (``framework.utils.monitoring.DeviceSwitch`` = csrc/switch.hip: the
"prior static" series, its median / trend and the two-state machine advanced by one small launch per step).  The
dynamic model's forward pass is then launched EVERY step with the machine's state as the predicate of its convolutions
(they return at once while the switch is static: ~1 ms of empty launches instead of a 7 ms pass), the prior is picked
by a select kernel, and nothing the step needs ever visits the host: no blocking read-back inside a step, and with
several ranks no host between the switch scalars' all-reduce and the decision.  The host-side ``Monitor`` keeps
receiving every sample (asynchronously) and only feeds the log; ``model_select.current`` reads the device state when
somebody looks at it.  ``ONDA_DEVICE_SWITCH=0`` (or a CPU device) keeps the decision on the host, as the reference has it.
"""
import os

import torch

from onda_amd import dist as odist
from onda_amd import ops
from onda_amd._lib import query
from onda_amd.config import unset
from onda_amd.framework.domain_adaptation.methods.prototypes import online_proDA
from onda_amd.framework.utils.monitoring import DeviceSwitch

DEVICE_SWITCH = os.environ.get("ONDA_DEVICE_SWITCH", "1") != "0"


class model_select:
    """Two-state switch: outside the gray area the static prior's (median) confidence decides; inside it, the last
    significant trend of that confidence does (fixture G5)."""
    static = 0
    dynamic = 1

    def __init__(self, start=0, gray_area=(0.84, 0.88), dev_threshold=0.0002) -> None:
        self._current = start
        self._current_dev = start
        self.freeze = False
        self.gray_area = gray_area
        self.dev_threshold = dev_threshold
        self.device_switch = None  # a DeviceSwitch: the state then lives on the GPU and is read from there on demand

    def _state(self, index, host_value):
        if self.device_switch is None:
            return host_value
        return int(self.device_switch.istate[index].item())  # (blocking: for whoever looks between steps)

    @property
    def current(self):
        return self._state(2, self._current)

    @current.setter
    def current(self, value):
        self._current = value
        if self.device_switch is not None:
            self.device_switch.istate[2] = int(value)
            self.device_switch.flag[0] = int(value)

    @property
    def current_dev(self):
        return self._state(3, self._current_dev)

    @current_dev.setter
    def current_dev(self, value):
        self._current_dev = value
        if self.device_switch is not None:
            self.device_switch.istate[3] = int(value)

    def eval(self):
        self.freeze = True

    def train(self):
        self.freeze = False

    def evaluate(self, confidence, dev_value):
        if self.freeze:
            return
        if dev_value > self.dev_threshold:  # a significant trend is remembered across steps (the reference's two tests, :24-27)
            self.current_dev = self.static
        elif dev_value < -self.dev_threshold:
            self.current_dev = self.dynamic
        low, high = self.gray_area[0], self.gray_area[1]
        if confidence < low:
            self.current = self.dynamic
        elif confidence > high:
            self.current = self.static
        else:
            self.current = self.current_dev


class hybrid_proDA(online_proDA):
    def __init__(self, model, cfg, cfg_spec) -> None:
        self.model_select = model_select(model_select.static, cfg_spec.GRAY_AREA, cfg_spec.DEV_THRESH)
        super().__init__(model, cfg, cfg_spec)
        self._dsw = None
        if DEVICE_SWITCH and torch.device(self.device).type == "cuda" and cfg_spec.STATIC_LAMBDA > 0 and self.intensity_ma.limit \
                and self.intensity_ma.limit <= query("onda_switch_max_window"):  # (longer windows: the host-side switch)
            spec = cfg_spec
            smoothed = not unset(spec.EXP_PR_STATIC) and bool(spec.EXP_PR_STATIC)
            monitor = self.intensity_ma  # the device series takes the host monitor's own settings (window, constant, trend)
            self._dsw = DeviceSwitch(self.device, monitor.limit, monitor.exp_const, monitor.dev_func, spec.GRAY_AREA,
                                     spec.DEV_THRESH, model_select.static, smoothed)
            self.model_select.device_switch = self._dsw
            self.intensity_ma.gated.add("prior dynamic")

    def _concurrent_ok(self, n_shards):
        # the decision stays on the device and the dynamic pass is predicated: nothing between the static model's pass and
        # the dynamic model's waits for the host, so the whole chain can sit on a side stream
        return self._dsw is not None and n_shards == 1 and ops.predicates_supported() and not self.intensity_ma.freeze

    # ---- the switch on the device ------------------------------------------------------------------------------------
    def _switch_scalars(self, ts):
        if self._dsw is None:
            return super()._switch_scalars(ts)
        rows = [torch.stack([t["conf_ema"], t["conf_static"]]) for t in ts]
        vals = rows[0] if len(rows) == 1 else torch.stack(rows).mean(0)
        if not self.intensity_ma.freeze:
            odist.all_reduce_mean(vals)            # one decision for all ranks, still on the device
            if not self.model_select.freeze:
                self._dsw.step(vals[1])            # "prior static" into the device ring; the machine moves; flag is set
        return vals                                # (no fetch function: nothing travels to the host for the decision)

    def _record_switch_scalars(self, t, fetch):
        if self._dsw is None:
            return super()._record_switch_scalars(t, fetch)
        # the host-side monitor gets the two samples for the log, by an asynchronous copy nobody waits for
        self.intensity_ma.add_device(["prior EMA", "prior static"], fetch)

    def _mixed_prior(self, t, image, deferred):
        if self._dsw is None:
            return super()._mixed_prior(t, image, deferred)
        lam = self.cfg_spec.DYNAMIC_LAMBDA
        if lam <= 0:
            return t["prior"]
        flag = self._dsw.flag
        if ops.predicates_supported():
            with ops.predicated(flag):             # the pass is launched either way; its convolutions obey the flag
                _, prior_dynamic, conf_dyn, _ = self._forward_prior(self.dynamic_model, image)
        elif int(flag.item()):                     # (conv modes without predicates: the host reads the device's decision)
            _, prior_dynamic, conf_dyn, _ = self._forward_prior(self.dynamic_model, image)
        else:
            return t["prior"]
        deferred.put("prior dynamic", ops.gate_scalar(flag, conf_dyn))
        return ops.select_prior(flag, t["prior"], 1.0, prior_dynamic, lam)

    def _prior_plan(self):
        """The static prior, unless the switch is in its dynamic state: then the dynamic model's prior REPLACES it
        (reference :57-75)."""
        spec, monitor = self.cfg_spec, self.intensity_ma
        smoothed = not unset(spec.EXP_PR_STATIC) and spec.EXP_PR_STATIC
        confidence = monitor.exp("prior static") if smoothed else monitor.avg("prior static")
        self.model_select.evaluate(confidence, monitor.dev_avg("prior static"))
        if self.model_select.current == model_select.dynamic and spec.DYNAMIC_LAMBDA > 0:
            return 0.0, spec.DYNAMIC_LAMBDA
        return 1.0, 0.0

    def prototype_predictions(self, batch):
        if "label" not in batch:
            batch["label"] = 0
        return super().prototype_predictions(batch)

    def models_eval(self):
        self.model_select.eval()
        return super().models_eval()

    def models_default_config(self):
        self.model_select.train()
        return super().models_default_config()
