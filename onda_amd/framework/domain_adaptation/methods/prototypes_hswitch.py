"""Drop-in for ``framework/domain_adaptation/methods/prototypes_hswitch.py``: ``hswitch_proDA``, the confidence
switch of ``configs/confidence_switch.yml`` (reference :27-84).  Only the prior mixing differs from ``online_proDA``."""
from onda_amd.framework.domain_adaptation.methods.prototypes import online_proDA, switch_batch_statistics  # noqa: F401


class hswitch_proDA(online_proDA):
    def _prior_plan(self):
        """prior = s * (teacher/static prior) + (1 - s) * DYNAMIC_LAMBDA * dynamic prior, with s the share of the static
        model: a linear ramp of the static prior's median confidence from 0.82 to 0.94 (SOFT_TRANS) or a hard
        threshold at SWITCH_PRIOR_THRESH."""
        spec = self.cfg_spec
        confidence = self.intensity_ma.avg("prior static")
        if spec.SOFT_TRANS:
            share = max(min(confidence * (25.0 / 3) - (41.0 / 6), 1), 0)
        else:
            share = int(confidence > spec.SWITCH_PRIOR_THRESH)
        self.intensity_ma.add({"percentage_static": share})
        if spec.DYNAMIC_LAMBDA > 0 and share < 1:
            return share, (1 - share) * spec.DYNAMIC_LAMBDA
        return share, 0.0
