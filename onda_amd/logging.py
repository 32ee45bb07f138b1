"""Where the mirrored training loops send their log dictionaries.

The reference calls ``wandb.log(dict)`` (prototypes.py:482,519; segmentation.py:92,137).  Here: ``wandb.log`` when
``wandb`` is importable and a run is active (so the unchanged driver behaves as with the reference), otherwise the
sink installed with ``set_sink`` (tests, bench), otherwise nothing.  Device scalars are read back in one transfer.
"""
import torch

_sink = None


def set_sink(fn):
    """fn(dict) receives every log dictionary (None: back to wandb-or-nothing)."""
    global _sink
    _sink = fn


def _host(d):
    keys = [k for k, v in d.items() if isinstance(v, torch.Tensor) and v.numel() == 1]
    if keys:
        vals = torch.stack([d[k].detach().reshape(()).float() for k in keys]).tolist()
        d = dict(d, **dict(zip(keys, vals)))
    return d


def log(d):
    if _sink is not None:
        return _sink(_host(d))
    try:
        import wandb
    except ImportError:
        return None
    if getattr(wandb, "run", None) is not None:
        wandb.log(_host(d))
    return None
