"""ORACLE (test infrastructure): the per-domain adaptation loop and the domain stream around it, on the CPU.

Parity status: PINNED by fixture G14 (the reference's ``hybrid_proDA.train`` run over two synthetic domains with
``update_cfg_spec`` between them, every ``wandb.log`` dictionary captured; tests/golden/make_golden.py::g14).

Follows, in order:
  the domain loop                      train_ouda.py:227-261  (cfg_spec.set_, ORDER_OPTIONS, SKIP_CALC |= f_domain)
  online_proDA.train                   framework/domain_adaptation/methods/prototypes.py:466-520
  online_proDA.evaluate_update_dynamic prototypes.py:396-405
  online_proDA.buffer_update           prototypes.py:452-464
  da_model.evaluate / evaluate_all     framework/domain_adaptation/methods/adaptation_model.py:127-179
  da_model.test_on_samples             adaptation_model.py:181-200 (+ evaluate.py:112-120: the class map of a sample)
  fast_hist / per_class_iu             framework/utils/func.py:77-85
The step itself is oracle/step.py (pinned by G7).  Only tests/, smoke() and bench.py's cpu leg may import this.
"""
import numpy as np
import torch

from . import model


def confusion(label, pred, n):
    """func.py:77-80: rows = ground truth, columns = prediction; labels outside [0, n) are dropped."""
    keep = (label >= 0) & (label < n)
    return np.bincount(n * label[keep].astype(int) + pred[keep], minlength=n ** 2).reshape(n, n)


def iou_per_class(hist):
    """func.py:83-85"""
    return np.diag(hist) / (hist.sum(1) + hist.sum(0) - np.diag(hist) + np.finfo(float).eps)


class OracleLoop:
    """`adapter`: an oracle.step.OracleAdapter; `size`: (H, W) of the images (the `interp` target); `options`: the
    loop-level settings of cfg / cfg_spec (EPOCHS, SOURCE_REPEAT, SKIP_CALC, AUTO_DYNAMIC, probability_per_step =
    PERC_FILL_PER_DOMAIN * REPLAY_BUFFER / BATCH_SIZE as prototypes.py:70-78 computes it, samples per validation set)."""

    def __init__(self, adapter, size, num_classes=19, **options):
        self.ad, self.size, self.n = adapter, tuple(size), num_classes
        self.opt = dict(EPOCHS=1, SOURCE_REPEAT=1, SKIP_CALC=False, AUTO_DYNAMIC=None, probability_per_step=0.0, samples=10)
        self.opt.update(options)
        self.dynamic_update_counter = 0
        self.have_prototypes = False

    # ---- evaluation (student, eval mode) ----------------------------------------------------------------------------
    @torch.no_grad()
    def _class_maps(self, images):
        out = model.forward(images, self.ad.student, model.BNMode(False))[1]["out"]
        return model.upsample_argmax(out, self.size)[1]

    def evaluate(self, loader):
        hist = 0
        for batch in loader:
            maps = self._class_maps(batch["image"])
            for pred, label in zip(maps, batch["label"]):
                hist = hist + confusion(label.numpy().flatten(), pred.numpy().flatten(), self.n)
        return iou_per_class(hist)

    def evaluate_all(self, validation_loaders):
        log = {}
        for name, loader in validation_loaders.items():
            iou = self.evaluate(loader)
            log[f"Val mIoU model of {name}"] = np.nanmean(iou)
            log[f"Val std IoU model of {name}"] = np.nanstd(iou)
        return log

    def test_on_samples(self, validation_loaders):
        log = {}
        for name, loader in validation_loaders.items():
            it = iter(loader)
            for i in range(self.opt["samples"]):
                sample = next(it)
                log[f"Condition {name} sample {i}"] = self._class_maps(sample["image"][:1])[0].to(torch.uint8).numpy()
        return log

    # ---- pieces of the loop ------------------------------------------------------------------------------------------
    def evaluate_update_dynamic(self):
        if not self.opt["AUTO_DYNAMIC"]:
            return
        self.dynamic_update_counter += 1
        if self.dynamic_update_counter > 500 and abs(self.ad.stats.dev_avg("prior static")) > self.ad.cfg["DEV_THRESH"]:
            self.ad.refresh_dynamic()
            self.dynamic_update_counter = 0

    def buffer_update(self, batch_target, probability, trainloader):
        updates = 0
        if probability > 0:
            b = batch_target["image"].shape[0]
            chosen = np.where(np.random.rand(b) < probability)[0]
            for index in chosen:
                stored = batch_target["stored_predictions"]
                if stored.dim() == 4:  # (the reference upsamples inside this loop; one chosen sample per batch in G14)
                    batch_target["stored_predictions"] = model.upsample_argmax(stored, self.size)[0].argmax(1)
                trainloader.add_from_batch(batch_target, index)
                updates += 1
        return updates

    # ---- online_proDA.train ------------------------------------------------------------------------------------------
    def train(self, trainloader, targetloader, validation_loaders, emit):
        ad, opt = self.ad, self.opt
        if not opt["AUTO_DYNAMIC"]:
            ad.refresh_dynamic()
        if not opt["SKIP_CALC"]:
            if not self.have_prototypes:
                ad.proto = ad.initial_prototypes(list(trainloader))
                self.have_prototypes = True
            emit(self.evaluate_all(validation_loaders))
        steps = opt["EPOCHS"] * len(targetloader)
        update_prob = opt["probability_per_step"] / steps
        sources, targets = iter(trainloader), iter(targetloader)
        for i in range(steps):
            src = []
            for _ in range(opt["SOURCE_REPEAT"]):
                try:
                    src.append(next(sources))
                except StopIteration:
                    sources = iter(trainloader)
                    src.append(next(sources))
            try:
                trg = next(targets)
            except StopIteration:
                targets = iter(targetloader)
                trg = next(targets)
            b = trg["image"].shape[0]
            masks = tuple(model.draw_drop_mask(b) for _ in range(3))  # source student, target student, teacher
            log = ad.step(src[0], trg, masks)
            bb, k, h, w = ad.last["out"].shape
            trg["stored_predictions"] = ad.last["soft"].reshape(bb, h, w, k).permute(0, 3, 1, 2)
            self.evaluate_update_dynamic()
            ad.update_ema()
            log["Total buffer updates"] = self.buffer_update(trg, update_prob, trainloader)
            if (i + 1) % len(targetloader) == 0:
                log.update(self.evaluate_all(validation_loaders))
                log.update(self.test_on_samples(validation_loaders))
            emit(log)


def run_domains(loop, trainloader, domains, validation_loaders, emit, order_options=None, between=None):
    """train_ouda.py:227-261 for a list of target loaders: later domains skip the prototype initialisation
    (SKIP_CALC |= f_domain), `order_options[d]` overrides loop options for domain d (SCHEME.ORDER_OPTIONS), `between(d)`
    runs before domain d (G14 pre-sets the 500-step counter there)."""
    first = True
    for d, loader in enumerate(domains):
        for key, value in (order_options or {}).get(d, {}).items():
            loop.opt[key] = value
        loop.opt["SKIP_CALC"] = bool(loop.opt["SKIP_CALC"]) or not first
        first = False
        if between is not None:
            between(d)
        loop.train(trainloader, loader, validation_loaders, emit)
