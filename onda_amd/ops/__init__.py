"""Host-side operator layer: torch tensors in, HIP kernels (through the C ABI) underneath.

PyTorch supplies device memory (the caching allocator owns every buffer, workspaces
included), the current HIP stream and the autograd graph; all arithmetic of the hot path
runs in ``libonda_hip.so``.  Activations are dense NHWC tensors ``[B, H, W, C]`` (or channel
slices of one, row stride ``ld``); parameters keep the reference's shapes (OIHW conv weights)
so state_dicts, ``deepcopy`` and optimizers behave exactly as with the reference modules.

Nothing here falls back to eager torch math: a missing library raises at first use.
"""
import sys
import types

from . import _state, conv, core, limbs, loss, norm, tables
from ._state import BN_EPS, GN_EPS, GN_GROUPS, HEAD_PAD, STEM_K

# The switches live in `_state` (CONV_MODE, H2_PATH, GRAD_READY, PROFILE, PREDICATE, ROW_GROUPS, LIMB_ONLY, SHARE_GRADS,
# FUSE_BN_FINALIZE): `ops.CONV_MODE` reads it, `ops.CONV_MODE = "f32"` writes it -- the submodules look there at call time.
_FLAGS = ("CONV_MODE", "H2_PATH", "GRAD_READY", "PROFILE", "PREDICATE", "ROW_GROUPS", "LIMB_ONLY", "SHARE_GRADS", "FUSE_BN_FINALIZE")

# everything else of the submodules under the names the single module had (tests, tools, bench.py and the framework mirror use
# `ops.<name>`, underscore names included); to replace a function that OTHER ops code calls, patch it in its submodule
for _m in (core, limbs, tables, conv, norm, loss):
    for _k, _v in vars(_m).items():
        if not _k.startswith("__") and _k not in _FLAGS and not isinstance(_v, types.ModuleType):
            globals()[_k] = _v
del _m, _k, _v


class _OpsModule(types.ModuleType):
    def __getattr__(self, name):
        if name in _FLAGS:
            return getattr(_state, name)
        raise AttributeError(f"module {self.__name__!r} has no attribute {name!r}")

    def __setattr__(self, name, value):
        if name in _FLAGS:
            setattr(_state, name, value)
        else:
            super().__setattr__(name, value)


sys.modules[__name__].__class__ = _OpsModule
