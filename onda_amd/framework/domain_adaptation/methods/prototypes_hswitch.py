"""Drop-in for ``framework/domain_adaptation/methods/prototypes_hswitch.py``: ``hswitch_proDA``,
the confidence switch (``configs/confidence_switch.yml``).  Same kernels as the hybrid method;
only the prior mixing differs (reference :27-84)."""
import torch

from onda_amd.framework.domain_adaptation.methods.prototypes import online_proDA, switch_batch_statistics  # noqa: F401


class hswitch_proDA(online_proDA):
    def prototype_predictions(self, batch):
        """prior = s * static + (1 - s) * dynamic, with s the share of the static model: a linear
        ramp of the median static confidence between 0.82 and 0.94 (SOFT_TRANS) or a hard
        threshold at SWITCH_PRIOR_THRESH."""
        with torch.no_grad():
            image, pred_ema, prior, cls_ema = self._teacher_and_static(batch)
            if self.cfg_spec.SOFT_TRANS:
                vl = self.intensity_ma.avg("prior static")
                percentage_static = max(min(vl * (25.0 / 3) - (41.0 / 6), 1), 0)
            else:
                percentage_static = int(self.intensity_ma.avg("prior static") > self.cfg_spec.SWITCH_PRIOR_THRESH)
            self.intensity_ma.add({"percentage_static": percentage_static})
            prior *= percentage_static
            if self.cfg_spec.DYNAMIC_LAMBDA > 0 and percentage_static < 1:
                prior += (1 - percentage_static) * self.cfg_spec.DYNAMIC_LAMBDA * self._dynamic_prior(image)
        return self._labels_from(pred_ema, prior, cls_ema)
