"""Drop-in for the hot-path functions of ``framework/utils/loss.py``: ``cross_entropy_2d``
(:16-45) and ``rce`` (:88-112), hard-label branches, on the fused HIP loss kernel."""
from onda_amd import ops


def cross_entropy_2d(predict, target, soft=False):
    """predict f32[n,c,h,w], target int[n,h,w]; mean CE over pixels with 0 <= target != 255
    (NaN when no pixel qualifies, like the reference)."""
    assert not target.requires_grad
    assert predict.dim() == 4
    assert target.dim() == 3 or target.dim() == 4
    assert predict.size(0) == target.size(0), f"{predict.size(0)} vs {target.size(0)}"
    assert predict.size(-2) == target.size(-2), f"{predict.size(-2)} vs {target.size(-2)}"
    assert predict.size(-1) == target.size(-1), f"{predict.size(-1)} vs {target.size(-1)}"
    if soft:
        raise NotImplementedError("onda_amd: soft-label CE (SOFT_LABELS) is not on the hybrid_switch hot path")
    return ops.seg_losses(predict, target, 1.0, 0.0, 0.0)[0]


def rce(pred, labels, device, soft=False):
    if soft:
        raise NotImplementedError("onda_amd: soft-label RCE (SOFT_LABELS) is not on the hybrid_switch hot path")
    return ops.seg_losses(pred, labels, 0.0, 1.0, 0.0)[0]
