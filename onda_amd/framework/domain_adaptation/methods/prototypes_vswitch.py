"""Drop-in for ``framework/domain_adaptation/methods/prototypes_vswitch.py``: ``vswitch_proDA``,
the confidence-derivative switch (``configs/confidence_der_switch.yml``; reference :5-89)."""
import torch

from onda_amd.framework.domain_adaptation.methods.prototypes import online_proDA


class model_select:
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
        if self.freeze:
            return
        # the sign of the windowed derivative of the static prior's confidence picks the model
        if dev_value > self.threshold:
            self.current = self.static
        elif dev_value < -self.threshold:
            self.current = self.dynamic


class vswitch_proDA(online_proDA):
    def __init__(self, model, cfg, cfg_spec) -> None:
        super().__init__(model, cfg, cfg_spec)
        self.model_select = model_select(model_select.static, cfg_spec.SWITCH_PRIOR_THRESH)

    def prototype_predictions(self, batch):
        with torch.no_grad():
            image, pred_ema, prior, cls_ema = self._teacher_and_static(batch)
            self.model_select.evaluate(self.intensity_ma.dev_avg("prior static"))
            if self.model_select.current == model_select.dynamic and self.cfg_spec.DYNAMIC_LAMBDA > 0:
                prior = self.cfg_spec.DYNAMIC_LAMBDA * self._dynamic_prior(image)
        return self._labels_from(pred_ema, prior, cls_ema)

    def models_eval(self):
        self.model_select.eval()
        return super().models_eval()

    def models_default_config(self):
        if hasattr(self, "model_select"):
            self.model_select.train()
        return super().models_default_config()
