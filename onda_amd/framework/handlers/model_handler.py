"""Drop-in for ``framework/handlers/model_handler.py``: ``get_model(cfg, n_classes)`` (:14-60)."""
import types

import torch

from onda_amd.framework.model.deeplabv2 import get_deeplab_v2

MODEL_NAMES = ["DeepLabv2-Resnet50", "DeepLabv2-Resnet101"]
_LAYERS = {"DeepLabv2-Resnet50": [3, 4, 6, 3], "DeepLabv2-Resnet101": [3, 4, 23, 3]}


def get_model(cfg, n_classes):
    assert cfg.MODEL.NAME in MODEL_NAMES, f"cfg.MODEL.NAME should be in {MODEL_NAMES} (the HIP path covers the DeepLabV2 ResNets)"
    # built multi_level=True like the reference, so the (never executed) layer5 head is part of the state_dict
    model = get_deeplab_v2(num_classes=n_classes, layers=_LAYERS[cfg.MODEL.NAME], multi_level=True,
                           classifier=cfg.MODEL.CLASSIFIER)
    load = cfg.MODEL.LOAD
    if load is not None and load != "None" and not (isinstance(load, dict) and not load):
        state = torch.load(load, map_location="cpu")
        if isinstance(state, types.MethodType):
            state = state()
        if "imagenet" in load.lower():
            merged = model.state_dict().copy()
            for key in state:
                parts = key.split(".")
                start = 1 if parts[0] in ("Scale", "module") else 0
                if parts[start] not in ("layer5", "fc"):
                    merged[".".join(parts[start:])] = state[key]
            model.load_state_dict(merged)
        else:
            model.load_state_dict(state)
    model.multi_level = cfg.MODEL.MULTI_LEVEL
    model.to(cfg.OTHERS.DEVICE)
    return model
