"""ORACLE (test infrastructure): the optimizer semantics the reference ends up with.

Parity status: PINNED by fixture G6 (torch.optim.SGD, for-loop implementation, fed the
reference's own parameter groups).

``ResNetMulti.optim_parameters`` (framework/model/deeplabv2.py:397-439) walks
``modules()`` x ``parameters()``, so every backbone conv weight is yielded once per
enclosing module: 3x for block convs, 4x for ``downsample.0.weight``, 1x for the stem.
torch.optim.SGD only warns; its for-loop implementation (the only one in the authors'
torch 1.7.1) then applies the update once per occurrence, sequentially
(adaptation_model.py:88-93).
"""
import torch


def backbone_multiplicity(name: str) -> int:
    """How many times `name` (a state_dict key of a trainable backbone tensor) occurs in
    parameter group 0."""
    if name == "conv1.weight":
        return 1
    if ".downsample.0." in name:
        return 4
    return 3


def param_groups(sd_names):
    """(group0 [(name, times)], group1 [name]).  Group 1 is layer6 only: `multi_level` is
    already False when the optimizer is built (model_handler.py:58 runs before
    adaptation_model.py:88), so get_10x_lr_params (deeplabv2.py:421-433) skips layer5."""
    g0, g1 = [], []
    for n in sd_names:
        if n.startswith("layer5."):
            continue
        if n.startswith("layer6."):
            g1.append(n)
        elif n.endswith("weight") and (".conv" in n or n == "conv1.weight" or ".downsample.0." in n):
            g0.append((n, backbone_multiplicity(n)))
    return g0, g1


@torch.no_grad()
def sgd_apply(p, g, buf, lr, times=1, momentum=0.9, weight_decay=1e-4, first_step="torch2"):
    """`times` sequential SGD updates of one tensor with the same gradient.
    `buf` is the momentum buffer or None before the first step; returns the buffer.

    On the very first step the two torch generations differ for a duplicated tensor:
    torch 1.7.1 creates the buffer at the first occurrence and accumulates into it at the
    later ones (first_step="torch1.7"); torch >= 2 collects the (missing) buffers before
    its loop, so every occurrence starts a fresh buffer and the last one is kept
    (first_step="torch2" -- what runs in this image and therefore what G6 pins)."""
    fresh = buf is None
    for _ in range(times):
        d = g + weight_decay * p
        if buf is None or (fresh and first_step == "torch2"):
            buf = d.clone()
        else:
            buf.mul_(momentum).add_(d)
        p.sub_(lr * buf)
    return buf


def lr_poly(base_lr, it, max_it, power):
    """framework/utils/func.py:45-47."""
    return base_lr * ((1 - float(it) / max_it) ** power)
