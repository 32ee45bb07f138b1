"""Drop-in for ``framework/domain_adaptation/methods/prototypes_hybrid_switch.py``: ``model_select`` (:5-34) and
``hybrid_proDA`` (:37-109), the hybrid static/dynamic switch of ``configs/hybrid_switch.yml``.

The method differs from ``online_proDA`` only in how the priors are mixed, so it only supplies ``_prior_plan``: the
step's machinery (pipelined teacher / static / student passes, deferred monitor entries, gradient exchange) is the
base class's.
"""
from onda_amd.config import unset
from onda_amd.framework.domain_adaptation.methods.prototypes import online_proDA


class model_select:
    """Two-state switch: outside the gray area the static prior's (median) confidence decides; inside it, the last
    significant trend of that confidence does (fixture G5)."""
    static = 0
    dynamic = 1

    def __init__(self, start=0, gray_area=(0.84, 0.88), dev_threshold=0.0002) -> None:
        self.current = start
        self.current_dev = start
        self.freeze = False
        self.gray_area = gray_area
        self.dev_threshold = dev_threshold

    def eval(self):
        self.freeze = True

    def train(self):
        self.freeze = False

    def evaluate(self, confidence, dev_value):
        if self.freeze:
            return
        if abs(dev_value) > self.dev_threshold:  # a significant trend is remembered across steps
            self.current_dev = self.static if dev_value > 0 else self.dynamic
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
