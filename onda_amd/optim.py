"""SGD with the semantics the reference actually runs, as one multi-tensor HIP launch.

The reference builds ``torch.optim.SGD(model.optim_parameters(lr), momentum, weight_decay)``
(``adaptation_model.py:88-93``).  ``optim_parameters`` yields every backbone conv weight once
per enclosing module (3x, 4x for the projection shortcuts; ``deeplabv2.py:397-439``) and
torch only warns about the duplicates, so its for-loop implementation applies the update
once per occurrence, sequentially.  ``ReplaySGD`` keeps the same param_groups (so
``optimizer.param_groups[i]["lr"]`` assignments work unchanged), de-duplicates internally
and replays ``times`` sequential updates per element inside the kernel.  The first step
follows torch >= 2 (every occurrence starts a fresh momentum buffer -- what the oracle is
pinned against, fixture G6).
"""
import warnings

import torch

from . import ops


class ReplaySGD(torch.optim.Optimizer):
    def __init__(self, params, lr, momentum=0.9, weight_decay=0.0):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")  # "parameter group contains duplicate parameters"
            super().__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay))

    flat_zero = None  # set by the multi-GPU gradient exchange: gradients are views of one flat buffer, zeroed in place
    grad_scale = 1.0  # the next step() multiplies the gradients by this (1 / world after an exchange that left rank SUMS)

    def zero_grad(self, set_to_none=True):
        if self.flat_zero is not None:
            return self.flat_zero()
        return super().zero_grad(set_to_none)

    @torch.no_grad()
    def step(self, closure=None):
        by_cfg = {}
        for group in self.param_groups:
            counts = {}
            order = []
            for p in group["params"]:
                if id(p) not in counts:
                    counts[id(p)] = 0
                    order.append(p)
                counts[id(p)] += 1
            for p in order:
                if p.grad is None:
                    continue
                st = self.state[p]
                fresh = "momentum_buffer" not in st
                if fresh:
                    st["momentum_buffer"] = torch.empty_like(p, memory_format=torch.contiguous_format)
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                key = (group["momentum"], group["weight_decay"])
                by_cfg.setdefault(key, []).append((p, g, st["momentum_buffer"], group["lr"], counts[id(p)], fresh))
        for (momentum, wd), items in by_cfg.items():
            ops.sgd_multi(items, momentum, wd, self.grad_scale)
        self.grad_scale = 1.0
