"""Drop-in for ``framework/domain_adaptation/methods/prototypes_vswitch.py``: ``vswitch_proDA``, the
confidence-derivative switch of ``configs/confidence_der_switch.yml`` (reference :5-89)."""
from onda_amd.framework.domain_adaptation.methods.prototypes import online_proDA


class model_select:
    """The sign of the windowed derivative of the static prior's confidence picks the model."""
    static = 0
    dynamic = 1

    def __init__(self, start=0, threshold_c=0.00028) -> None:
        self.current = start
        self.freeze = False
        self.threshold = threshold_c

    def eval(self):
        self.freeze = True

    def train(self):
        self.freeze = False

    def evaluate(self, dev_value):
        if self.freeze or abs(dev_value) <= self.threshold:
            return
        self.current = self.static if dev_value > 0 else self.dynamic


class vswitch_proDA(online_proDA):
    def __init__(self, model, cfg, cfg_spec) -> None:
        super().__init__(model, cfg, cfg_spec)
        self.model_select = model_select(model_select.static, cfg_spec.SWITCH_PRIOR_THRESH)

    def _prior_plan(self):
        self.model_select.evaluate(self.intensity_ma.dev_avg("prior static"))
        if self.model_select.current == model_select.dynamic and self.cfg_spec.DYNAMIC_LAMBDA > 0:
            return 0.0, self.cfg_spec.DYNAMIC_LAMBDA
        return 1.0, 0.0

    def models_eval(self):
        self.model_select.eval()
        return super().models_eval()

    def models_default_config(self):
        if hasattr(self, "model_select"):
            self.model_select.train()
        return super().models_default_config()
