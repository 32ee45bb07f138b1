"""Make ``import framework...`` resolve to the HIP-backed drop-in."""
import importlib
import os
import sys

_MODULES = [
    "framework", "framework.model", "framework.model.deeplabv2", "framework.handlers",
    "framework.handlers.model_handler", "framework.handlers.adaptation_method_handler",
    "framework.utils", "framework.utils.func", "framework.utils.loss", "framework.utils.monitoring",
    "framework.domain_adaptation", "framework.domain_adaptation.methods",
    "framework.domain_adaptation.methods.adaptation_model", "framework.domain_adaptation.methods.prototype_handler",
    "framework.domain_adaptation.methods.prototypes", "framework.domain_adaptation.methods.prototypes_hybrid_switch",
    "framework.domain_adaptation.methods.prototypes_hswitch", "framework.domain_adaptation.methods.prototypes_vswitch",
    "framework.domain_adaptation.methods.segmentation", "framework.dataset", "framework.dataset.buffer_db",
]


def install(reference_root=None):
    """Alias the mirrored modules under the reference's names.  With `reference_root`, package
    search paths are extended so that every module NOT mirrored here (datasets, configs, ...)
    is still found in the reference checkout."""
    for name in _MODULES:
        mod = importlib.import_module("onda_amd." + name)
        sys.modules[name] = mod
        if reference_root and hasattr(mod, "__path__"):
            extra = os.path.join(reference_root, *name.split("."))
            if os.path.isdir(extra) and extra not in mod.__path__:
                mod.__path__.append(extra)
    return sys.modules["framework"]
