"""Multi-GPU exchange for the batch-sharded adaptation step: one process per GPU,
``torch.distributed`` (backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).

Per step there are exactly three exchanges, all of them skipped when the
process group is absent or has one rank:
  1. gradients: ONE all-reduce of a flat 46 M-float bucket after the second backward
     (source + target gradients are accumulated locally first);
  2. prototype statistics [sum feat | sum feat^2 | count] (9 747 floats) before the EMA
     blend, so the blend equals the global-batch formula;
  3. the switch scalars (mean max-probabilities) so that every rank's ``model_select``
     takes the same static/dynamic branch.
BatchNorm statistics stay rank-local (each rank normalises over its own micro-batch, like
the bs=4 reference); teacher EMA and SGD are replicated (identical inputs, identical result).
"""
import os

import torch
import torch.distributed as dist


def is_on():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def world_size():
    return dist.get_world_size() if is_on() else 1


def rank():
    return dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0


def init_from_env(backend=None):
    """Initialise from torchrun's environment (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*).
    Returns (rank, world, local_rank); a no-op for a single process."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rk = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # test hooks: several ranks on ONE GPU over gloo exercise the whole multi-rank step logic on a
    # single-GPU box (ONDA_DIST_BACKEND=gloo ONDA_FORCE_DEVICE=0); production leaves both unset
    if "ONDA_FORCE_DEVICE" in os.environ:
        local = int(os.environ["ONDA_FORCE_DEVICE"])
    backend = backend or os.environ.get("ONDA_DIST_BACKEND")
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rk, world_size=world)
    return rk, world, local


def all_reduce_sum(t):
    if is_on():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def all_reduce_mean(t):
    if is_on():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        t /= dist.get_world_size()
    return t


def all_reduce_max(t):
    if is_on():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t


def barrier():
    if is_on():
        dist.barrier()


class GradSync:
    """Averages the gradients of `module` over ranks with one flat all-reduce."""

    def __init__(self, module):
        self.module = module

    def params_with_grad(self):
        seen, out = set(), []
        for p in self.module.parameters():
            if p.grad is not None and id(p) not in seen:
                seen.add(id(p))
                out.append(p)
        return out

    @torch.no_grad()
    def all_reduce(self):
        if not is_on():
            return 0
        params = self.params_with_grad()
        if not params:
            return 0
        grads = [p.grad for p in params]
        flat = torch.cat([g.reshape(-1) for g in grads])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat /= dist.get_world_size()
        torch._foreach_copy_(grads, [v.view_as(g) for v, g in zip(flat.split([g.numel() for g in grads]), grads)])
        return flat.numel()
