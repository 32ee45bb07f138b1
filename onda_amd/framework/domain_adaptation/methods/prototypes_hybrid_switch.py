"""Drop-in for ``framework/domain_adaptation/methods/prototypes_hybrid_switch.py``:
``model_select`` (:5-34) and ``hybrid_proDA`` (:37-109), the hybrid static/dynamic switch."""
import torch

from onda_amd.config import unset
from onda_amd.framework.domain_adaptation.methods.prototypes import online_proDA


class model_select:
    static = 0
    dynamic = 1

    def __init__(self, start=0, gray_area=(0.84, 0.88), dev_threshold=0.0002) -> None:
        self.current = start
        self.freeze = False
        self.current_dev = start
        self.gray_area = gray_area
        self.dev_threshold = dev_threshold

    def eval(self):
        self.freeze = True

    def train(self):
        self.freeze = False

    def evaluate(self, confidence, dev_value):
        if self.freeze:
            return
        # the trend of the static prior's confidence is remembered across steps ...
        if dev_value > self.dev_threshold:
            self.current_dev = self.static
        elif dev_value < -self.dev_threshold:
            self.current_dev = self.dynamic
        # ... and only decides inside the gray area
        lo, hi = self.gray_area[0], self.gray_area[1]
        self.current = self.dynamic if confidence < lo else self.static if confidence > hi else self.current_dev


class hybrid_proDA(online_proDA):
    def __init__(self, model, cfg, cfg_spec) -> None:
        self.model_select = model_select(model_select.static, cfg_spec.GRAY_AREA, cfg_spec.DEV_THRESH)
        super().__init__(model, cfg, cfg_spec)

    def prototype_predictions(self, batch):
        """Priors with the switch: the static prior is used unless the switch is in its dynamic
        state, in which case the dynamic model's prior REPLACES it (reference :45-101)."""
        with torch.no_grad():
            if "label" not in batch:
                batch["label"] = 0
            image, pred_ema, prior, cls_ema = self._teacher_and_static(batch)
            if not unset(self.cfg_spec.EXP_PR_STATIC) and self.cfg_spec.EXP_PR_STATIC:
                static_conf = self.intensity_ma.exp("prior static")
            else:
                static_conf = self.intensity_ma.avg("prior static")
            self.model_select.evaluate(static_conf, self.intensity_ma.dev_avg("prior static"))
            if self.model_select.current == model_select.dynamic and self.cfg_spec.DYNAMIC_LAMBDA > 0:
                prior = self.cfg_spec.DYNAMIC_LAMBDA * self._dynamic_prior(image)
        return self._labels_from(pred_ema, prior, cls_ema)

    def models_eval(self):
        self.model_select.eval()
        return super().models_eval()

    def models_default_config(self):
        self.model_select.train()
        return super().models_default_config()
