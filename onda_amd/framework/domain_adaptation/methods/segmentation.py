"""Drop-in for ``framework/domain_adaptation/methods/segmentation.py`` (``train``, :18-138): plain supervised
segmentation training -- BASELINE config 2: forward -> bilinear upsample (align_corners) to the label resolution ->
cross-entropy -> backward -> SGD with the duplicated parameter groups -> poly learning rate.

``train(model, train_loader, validation_loaders, cfg, cfg_spec)`` keeps the reference's signature and loop
(:62-100); the unit of work lives in ``SegmentationTrainer.step`` so that bench.py can time exactly it.  Every
forward/backward op is a HIP kernel; the loss is evaluated on the upsampled logits by the fused loss kernel;
validation uses the fused upsample -> argmax -> confusion-matrix kernel (adaptation_model.da_model.evaluate).
Not mirrored (outside SURVEY section 8): the DOMAIN_ANALYSIS branch, sample images, entropy logging, the
per-phase timers (PytorchSpeedMeasure).
"""
import os

import numpy as np
import torch

from onda_amd import dist as odist
from onda_amd import logging as olog
from onda_amd import ops
from onda_amd.config import unset
from onda_amd.framework.utils.func import lr_poly, per_class_iu
from onda_amd.optim import ReplaySGD


class SegmentationTrainer:
    def __init__(self, model, cfg, cfg_spec):
        if not unset(cfg.DOMAIN_ANALYSIS):
            raise NotImplementedError("onda_amd: the DOMAIN_ANALYSIS variant of segmentation.train is outside the hot path")
        self.model, self.cfg, self.spec = model, cfg, cfg_spec
        self.device = cfg.OTHERS.DEVICE
        width, height = cfg.SCHEME.RESOLUTION
        self.size = (height, width)
        self.base_lr = cfg_spec.LEARNING_RATE
        self.optimizer = ReplaySGD(model.optim_parameters(self.base_lr), lr=self.base_lr, momentum=cfg_spec.MOMENTUM,
                                   weight_decay=cfg_spec.WEIGHT_DECAY)
        self.steps_done = 0
        # several ranks: batch-sharded data parallel -- the same flat-buffer gradient exchange as the adaptation step
        # (buckets leave while the backward pass is still running; BatchNorm batch statistics stay rank-local)
        previous = model.__dict__.get("_onda_grad_sync")
        if previous is not None:
            previous.close()
        skip = (lambda name: name.startswith("layer5.")) if not getattr(model, "multi_level", True) else None
        self._float_buffers = [b for b in model.buffers() if b.dtype == torch.float32]  # BatchNorm running statistics
        self._grad_sync = odist.GradSync(model, tail_floats=sum(b.numel() for b in self._float_buffers), skip=skip)
        model.__dict__["_onda_grad_sync"] = self._grad_sync
        if self._grad_sync.active:
            ops.GRAD_READY = self._grad_sync.grad_ready
            self.optimizer.flat_zero = self._grad_sync.zero

    def loss(self, batch):
        """loss_calc(interp(prediction), label) (+ 0.1 x the auxiliary head's when multi_level), segmentation.py:70-80."""
        aux, pred = self.model(batch["image"].to(self.device))
        label = batch["label"].to(self.device)
        total = None
        for weight, p in ((1.0, pred), (0.1, aux)):
            if p is None:
                continue
            logits = p["out"] if isinstance(p, dict) else p
            if tuple(label.shape[1:]) != tuple(self.size):
                raise RuntimeError(f"onda_amd: labels of {tuple(label.shape[1:])} for SCHEME.RESOLUTION {self.size}")
            part = ops.upsample_ce(logits, label)  # interp -> cross-entropy without the upsampled tensor
            total = weight * part if total is None else total + weight * part
        return total

    def adjust_learning_rate(self, total_steps):
        """_adjust_learning_rate (func.py:50-58): poly schedule; the second group runs at 10x."""
        lr = lr_poly(self.base_lr, self.steps_done, total_steps, self.spec.POWER)
        self.optimizer.param_groups[0]["lr"] = lr
        if len(self.optimizer.param_groups) > 1:
            self.optimizer.param_groups[1]["lr"] = lr * 10

    def step(self, batch, total_steps=None):
        """One training step (segmentation.py:66-88); returns the loss as a device scalar."""
        self.optimizer.zero_grad()
        loss = self.loss(batch)
        self._grad_sync.arm()
        loss.backward()
        if self._grad_sync.active:
            # the running statistics (each rank's own batch moved them) ride in the tail of the gradient buffer and come
            # back as rank means: replicas keep identical buffers, so validation and checkpoints do not depend on the rank
            bufs, tail = self._float_buffers, self._grad_sync.tail
            if bufs:
                torch.cat([b.reshape(-1) for b in bufs], out=tail[:tail.numel()])
            self._grad_sync.finish(mean=False)  # rank sums; the optimizer divides on the way in
            world = odist.world_size()
            self.optimizer.grad_scale = 1.0 / world
            if bufs:
                mean = tail / world
                torch._foreach_copy_(bufs, [v.view_as(b) for v, b in zip(mean.split([b.numel() for b in bufs]), bufs)])
                for b in bufs:
                    torch.autograd.graph.increment_version(b)
        self.optimizer.step()
        if total_steps:
            self.adjust_learning_rate(total_steps)
        self.steps_done += 1
        return loss.detach()

    def evaluate(self, loader):
        self.model.eval()
        n = self.cfg.NUM_CLASSES
        hist = torch.zeros(n, n, dtype=torch.int64, device=self.device)
        with torch.no_grad():
            for batch in loader:
                out = self.model(batch["image"].to(self.device))[1]
                ops.upsample_argmax_hist(out["out"] if isinstance(out, dict) else out, batch["label"], hist, n)
        self.model.train()
        return per_class_iu(hist.cpu().numpy())


def save_model(model, epoch, cfg):
    """``model_train_<source set>.pth`` under the snapshot directory, overwritten every epoch (reference :141-151: the file
    later runs name as MODEL.LOAD / LOAD_MODEL)."""
    root, set_ = cfg.SNAPSHOT_DIR, None
    if unset(root):
        root, set_ = cfg.OTHERS.SNAPSHOT_DIR, cfg.SCHEME.SOURCE
    else:
        set_ = cfg.DOMAIN_ANALYSIS.DATASET.TRAIN
    if odist.rank() == 0:  # the replicas are identical; several ranks writing one path at the same time tear the file
        os.makedirs(root, exist_ok=True)
        torch.save(model.state_dict(), os.path.join(root, f"model_train_{set_}.pth"))
    odist.barrier()


def train(model, train_loader, validation_loaders, cfg, cfg_spec=None):
    trainer = SegmentationTrainer(model, cfg, cfg_spec)
    loader = next(iter(train_loader.values())) if isinstance(train_loader, dict) else train_loader
    total = len(loader) * cfg_spec.EPOCHS
    pending = []
    for epoch in range(cfg_spec.EPOCHS):
        model.train()
        for batch in loader:
            pending.append(trainer.step(batch, total))
            if (trainer.steps_done - 1) % 10 == 0:  # the reference logs the running mean every 10 steps (:90-97)
                olog.log({"Segmentation loss": torch.stack(pending).mean(), "learning_rate": trainer.optimizer.param_groups[0]["lr"]})
                pending = []
        log = {"epoch": epoch}
        for name, val_loader in (validation_loaders or {}).items():
            iou = trainer.evaluate(val_loader)
            log[f"Val mIoU of {name}"] = np.nanmean(iou)
            log[f"Val std IoU of {name}"] = np.nanstd(iou)
        olog.log(log)
        save_model(model, epoch, cfg)
    return trainer
