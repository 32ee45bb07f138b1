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


def switch_batch_statistics(model, setting, batchnorms=None):
    """Freeze / unfreeze the running-statistics update of every BatchNorm2d of `model` (`batchnorms`: the list of them,
    when the caller keeps one -- the walk over the module tree is the cost of this function)."""
    assert isinstance(setting, bool), f"setting value should be a boolean, given: {setting}"
    for m in (model.modules() if batchnorms is None else batchnorms):
        if isinstance(m, nn.BatchNorm2d):
            m.track_running_stats = setting


def _batchnorms(model):
    return [(name, m) for name, m in model.named_modules() if isinstance(m, nn.BatchNorm2d)]


class batchnorm_stats:
    """A second set of BatchNorm states next to the live one (BN_POLICY "double": source and target statistics are
    swapped around the source pass).  Same methods as the reference's class (:39-72)."""

    def __init__(self, model) -> None:
        self.model = model
        self.memory = {}
        self.save()

    def save(self):
        self.memory = {name: deepcopy(m.state_dict()) for name, m in _batchnorms(self.model)}

    def load(self):
        for name, m in _batchnorms(self.model):
            m.load_state_dict(self.memory[name])

    def exchange(self):
        live = {name: deepcopy(m.state_dict()) for name, m in _batchnorms(self.model)}
        self.load()
        self.memory = live

    def compare(self):
        for name, m in _batchnorms(self.model):
            kept = self.memory[name]["running_mean"]
            print({"model": m.running_mean, "memory": kept, "diff": m.running_mean - kept})


class _Interp(nn.Module):
    """nn.Upsample(size, mode="bilinear", align_corners=True) on the HIP kernel."""

    def __init__(self, size):
        super().__init__()
        self.size = tuple(size)

    def forward(self, x):
        return ops.UpsampleFn.apply(x, self.size)


class da_model:
    def __init__(self, model, cfg, cfg_spec) -> None:
        self.model, self.cfg, self.cfg_spec = model, cfg, cfg_spec
        self.device = cfg.OTHERS.DEVICE
        self.bn = batchnorm_stats(model)
        if not (isinstance(cfg.OTHERS.ECE_SKIP, bool) and cfg.OTHERS.ECE_SKIP):
            raise NotImplementedError("onda_amd: ECE recording is outside the hot path; set OTHERS.ECE_SKIP: True "
                                      "(as hybrid_switch.yml / static_model.yml do)")
        self.ece_record = False
        # the reference's torch.optim.SGD(model.optim_parameters(lr), ...) (:88-93): same parameter groups, duplicates
        # included; ReplaySGD replays the for-loop semantics of those duplicates in one HIP launch
        self.optimizer = ReplaySGD(model.optim_parameters(cfg_spec.LEARNING_RATE), lr=cfg_spec.LEARNING_RATE,
                                   momentum=cfg_spec.MOMENTUM, weight_decay=cfg_spec.WEIGHT_DECAY)
        width, height = cfg.SCHEME.RESOLUTION
        self.interp = _Interp((height, width))
        self.eval_metric_list = []
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
        """Poly schedule, scaled per parameter group by MODEL.LR_RATIO ("80:10" in hybrid_switch.yml; reference :116-125)."""
        if unset(self.cfg.MODEL.LR_RATIO):
            self.cfg.MODEL.LR_RATIO = "1:10"
        base = lr_poly(self.cfg_spec.LEARNING_RATE, step, total_steps, self.cfg_spec.POWER)
        for group, ratio in zip(self.optimizer.param_groups, self.cfg.MODEL.LR_RATIO.split(":")):
            group["lr"] = base * int(ratio)

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
        """{"Val mIoU <predictor> of <set>", "Val std IoU ..."} for every validation set (+ the pending extra metrics)."""
        report = {}
        for set_name, loader in validation_loaders.items():
            for predictor, iou in self.evaluate(loader).items():
                report[f"Val mIoU {predictor} of {set_name}"] = np.nanmean(iou)
                report[f"Val std IoU {predictor} of {set_name}"] = np.nanstd(iou)
            report.update({f"{metric} {set_name}": value for metric, value in self.eval_metric_list})
            self.eval_metric_list = []
        return report

    def test_on_samples(self, validation_loaders, count=10):
        """Class maps of the first `count` samples of every validation set (reference :181-200, which wraps each into a
        wandb image through ``segment_sample``, evaluate.py:112-120; wandb is outside the hot path): the map comes from the
        fused upsample -> argmax kernel, the entry is {"image", "prediction" u8[H,W], "label", "caption"}."""
        self.models_eval()
        log = {}
        width, height = self.cfg.SCHEME.RESOLUTION
        with torch.no_grad():
            for set_name, loader in validation_loaders.items():
                for i, sample in zip(range(count), loader):  # (a set with fewer samples gives what it has)
                    image, label = sample["image"][0], sample["label"][0]
                    out = self.model(image.unsqueeze(0).to(self.device))[1]
                    out = out["out"] if isinstance(out, dict) else out
                    log[f"Condition {set_name} sample {i}"] = {
                        "image": image.cpu().numpy(), "prediction": ops.upsample_argmax(out, (height, width))[0].cpu().numpy(),
                        "label": label.cpu().numpy(), "caption": f"Sample from {set_name}"}
        self.models_default_config()
        return log

    def save_prediction(self, prediction):
        """One file per batch under PREDICTION_SAVE/<set>/batch-<n>.pt (reference :222-235)."""
        key = self.cfg_spec.set_
        base_path = os.path.join(self.cfg_spec.PREDICTION_SAVE, "_".join(str(key)))
        if key not in self.prediction_counter:
            self.prediction_counter[key] = 0
            os.makedirs(base_path, exist_ok=True)
        torch.save(prediction, os.path.join(base_path, f"batch-{self.prediction_counter[key]}.pt"))
        self.prediction_counter[key] += 1

    def run_predictions(self, trg_loader, log_fn=None):
        """Student logits of every target batch to disk, with the mean max-probability logged (reference :237-250; the
        log goes to `log_fn` or the logging sink instead of wandb.log)."""
        from onda_amd import logging as olog
        emit = log_fn or olog.log
        self.models_eval()
        with torch.no_grad():
            for i, batch in enumerate(trg_loader):
                out = self.model(batch["image"].to(self.device))[1]["out"]
                confidence = out.softmax(dim=1).max(dim=1)[0].mean()
                emit({"Prediction confidence": confidence, "Progress": i * 100.0 / len(trg_loader)})
                self.save_prediction(out.cpu())
        self.models_default_config()

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
    """Evaluation-only wrapper: loads the newest ``*.pth`` of OTHERS.SNAPSHOT_DIR (unless it is "NONE")."""

    def __init__(self, model, cfg, cfg_spec) -> None:
        super().__init__(model, cfg, cfg_spec)
        folder = self.cfg.OTHERS.SNAPSHOT_DIR
        if folder != "NONE":
            checkpoints = [p for p in Path(folder).iterdir() if "pth" in str(p)]
            super().load_model(max(checkpoints, key=os.path.getmtime))

    def models_eval(self):
        self.model.eval()

    def models_default_config(self):
        self.model.eval()
