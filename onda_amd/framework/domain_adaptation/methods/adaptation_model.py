"""Drop-in for the hot-path part of ``framework/domain_adaptation/methods/adaptation_model.py``:
``switch_batch_statistics`` (:29-36), ``batchnorm_stats`` (:39-72), ``da_model`` (:75-250:
optimizer construction, LR schedule, evaluation, checkpoints) and ``evaluation`` (:252-265).

Evaluation computes the class map with the fused upsample->argmax kernel (no 159 MB
upsampled tensor); the confusion matrix stays the reference's numpy bincount.
"""
import abc
import os
from copy import deepcopy
from pathlib import Path

import numpy as np
import torch
from torch import nn

from onda_amd import ops
from onda_amd.config import unset
from onda_amd.optim import ReplaySGD
from onda_amd.framework.utils.func import lr_poly, per_class_iu


def switch_batch_statistics(model, setting):
    """Freeze / unfreeze the running-statistics update of every BatchNorm2d of `model`."""
    assert isinstance(setting, bool), f"setting value should be a boolean, given: {setting}"
    for m in model.modules():
        if isinstance(m, nn.BatchNorm2d):
            m.track_running_stats = setting


class batchnorm_stats:
    def __init__(self, model) -> None:
        self.memory = {}
        self.model = model
        self.save()

    def _bn(self):
        return ((n, m) for n, m in self.model.named_modules() if isinstance(m, nn.BatchNorm2d))

    def save(self):
        for name, module in self._bn():
            self.memory[name] = deepcopy(module.state_dict())

    def load(self):
        for name, module in self._bn():
            module.load_state_dict(self.memory[name])

    def exchange(self):
        for name, module in self._bn():
            current = deepcopy(module.state_dict())
            module.load_state_dict(self.memory[name])
            self.memory[name] = current

    def compare(self):
        for name, module in self._bn():
            print({"model": module.running_mean, "memory": self.memory[name]["running_mean"],
                   "diff": module.running_mean - self.memory[name]["running_mean"]})


class _Interp(nn.Module):
    """nn.Upsample(size, mode="bilinear", align_corners=True) on the HIP kernel."""

    def __init__(self, size):
        super().__init__()
        self.size = tuple(size)

    def forward(self, x):
        return ops.UpsampleFn.apply(x, self.size)


class da_model:
    def __init__(self, model, cfg, cfg_spec) -> None:
        self.model = model
        self.bn = batchnorm_stats(model)
        self.cfg = cfg
        self.cfg_spec = cfg_spec
        self.device = cfg.OTHERS.DEVICE
        input_size_source = cfg.SCHEME.RESOLUTION
        learning_rate = cfg_spec.LEARNING_RATE
        # same parameter groups (duplicates included) and hyper-parameters as the reference's
        # torch.optim.SGD(...); ReplaySGD replays its for-loop semantics in one HIP launch
        self.optimizer = ReplaySGD(model.optim_parameters(learning_rate), lr=learning_rate,
                                   momentum=cfg_spec.MOMENTUM, weight_decay=cfg_spec.WEIGHT_DECAY)
        self.interp = _Interp((input_size_source[1], input_size_source[0]))
        self.eval_metric_list = []
        self.ece_record = not (isinstance(cfg.OTHERS.ECE_SKIP, bool) and cfg.OTHERS.ECE_SKIP)
        if self.ece_record:
            raise NotImplementedError("onda_amd: ECE recording is outside the hot path; set OTHERS.ECE_SKIP: True "
                                      "(as hybrid_switch.yml / static_model.yml do)")
        self.prediction_counter = {}

    @abc.abstractmethod
    def models_eval(self):
        pass

    @abc.abstractmethod
    def models_default_config(self):
        pass

    def update_cfg_spec(self, new_cfg):
        self.cfg_spec = new_cfg

    def adjust_learning_rate(self, step, total_steps):
        if unset(self.cfg.MODEL.LR_RATIO):
            self.cfg.MODEL.LR_RATIO = "1:10"
        ratios = [int(v) for v in self.cfg.MODEL.LR_RATIO.split(":")]
        learning_rate = lr_poly(self.cfg_spec.LEARNING_RATE, step, total_steps, self.cfg_spec.POWER)
        self.optimizer.param_groups[0]["lr"] = learning_rate * ratios[0]
        if len(self.optimizer.param_groups) > 1:
            self.optimizer.param_groups[1]["lr"] = learning_rate * ratios[1]

    def evaluate(self, validation_loader, additional_func={}):
        """mIoU of the student (and of any extra prediction function) over a loader."""
        function_dict = {"model": lambda x: self.model(x["image"].to(self.device))[1]["out"]}
        function_dict.update(additional_func)
        self.models_eval()
        n = self.cfg.NUM_CLASSES
        # the confusion matrices live on the GPU; one read-back per evaluation instead of a class
        # map per image (the fused kernel upsamples, takes the argmax and bins against the labels)
        counters = {key: torch.zeros(n, n, dtype=torch.int64, device=self.device) for key in function_dict}
        with torch.no_grad():
            for batch in validation_loader:
                for key, func in function_dict.items():
                    ops.upsample_argmax_hist(func(batch), batch["label"], counters[key], n)
        self.models_default_config()
        return {key: per_class_iu(count.cpu().numpy()) for key, count in counters.items()}

    def evaluate_all(self, validation_loaders):
        validation_log = {}
        for val_set, val_loader in validation_loaders.items():
            for key, value in self.evaluate(val_loader).items():
                validation_log[f"Val mIoU {key} of {val_set}"] = np.nanmean(value)
                validation_log[f"Val std IoU {key} of {val_set}"] = np.nanstd(value)
            for name, value in self.eval_metric_list:
                validation_log[f"{name} {val_set}"] = value
            self.eval_metric_list = []
        return validation_log

    def save_model(self, model_dict=None, prefix=""):
        if model_dict is None:
            model_dict = {"model": self.model}
        root = self.cfg.OTHERS.SNAPSHOT_DIR
        os.makedirs(root, exist_ok=True)
        for key, model in model_dict.items():
            torch.save(model.state_dict(), os.path.join(root, f"{key}_{prefix}.pth"))

    def load_model(self, path):
        print(f"Model {path} is being loaded")
        self.model.load_state_dict(torch.load(path))


class evaluation(da_model):
    def __init__(self, model, cfg, cfg_spec) -> None:
        super().__init__(model, cfg, cfg_spec)
        dirpath = self.cfg.OTHERS.SNAPSHOT_DIR
        if dirpath != "NONE":
            paths = sorted(Path(dirpath).iterdir(), reverse=True, key=os.path.getmtime)
            super().load_model([p for p in paths if "pth" in str(p)][0])

    def models_eval(self):
        self.model.eval()

    def models_default_config(self):
        self.model.eval()
