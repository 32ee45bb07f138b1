"""ORACLE (test infrastructure): one complete hybrid-switch adaptation step on the CPU.

Parity status: PINNED by fixture G7 (log dict, post-step weights and prototypes of the
reference's ``hybrid_proDA.step`` + ``update_ema`` on a 128x64, B=2 synthetic state).

Follows, in order:
  online_proDA.step                 framework/domain_adaptation/methods/prototypes.py:418-450
  online_proDA.supervised_loss      prototypes.py:157-189
  online_proDA.pseudolabel_loss     prototypes.py:275-372
  hybrid_proDA.prototype_predictions  prototypes_hybrid_switch.py:45-101
  online_proDA.update_ema           prototypes.py:407-416
with the hyper-parameters of configs/hybrid_switch.yml:25-62 as defaults.
"""
from copy import deepcopy

import torch

from . import losses, model, monitor, optim, prototypes

DEFAULTS = dict(
    LEARNING_RATE=1e-5, LR_RATIO=(80, 10), MOMENTUM=0.9, WEIGHT_DECAY=1e-4, POWER=0,
    EMA_UPDATE=0.999, EMA_LAMBDA=0.0, STATIC_LAMBDA=1.0, DYNAMIC_LAMBDA=1.0,
    MA_LAMBDA=0.9995, TAU=1.0, PSEUDO_THRESH=0.3, DISTANCE_MEASURE="mahalanobis",
    RCE_ALPHA=0.1, RCE_BETA=1.0, REGULARIZER_WEIGHT=0.1, BUFF_CE=1.0,
    GRAY_AREA=(0.83, 0.9), DEV_THRESH=2e-4, AVG_MONITOR_SIZE=200, EXP_MONITOR_CONST=0.003,
    CONF_REG_THRESH=1.0,
)


def _is_trainable(name, t):
    if not t.is_floating_point() or "running_" in name:
        return False
    # BN affine parameters are frozen (deeplabv2.py:26-48, :285-287, :360-362)
    is_bn = (".bn" in name or name.startswith("bn1.") or ".downsample.1." in name)
    return not is_bn


class OracleAdapter:
    """method: "hybrid" (prototypes_hybrid_switch.py:45-101, the BASELINE path), "online"
    (prototypes.py:208-273, used by static_model.yml / dynamic_model.yml), "hswitch"
    (prototypes_hswitch.py:27-84, confidence switch with soft transition) or "vswitch"
    (prototypes_vswitch.py:20-89, confidence-derivative switch).  They differ only in how the
    static / dynamic priors are mixed; fixtures G7 (hybrid) and G8 (the others) pin them."""

    def __init__(self, sd, proto_state, cfg=None, first_step="torch2", method="hybrid"):
        self.method = method
        self.cfg = dict(DEFAULTS, **(cfg or {}))
        self.student = {k: v.clone() for k, v in sd.items()}
        self.ema = deepcopy(self.student)
        self.dynamic = deepcopy(self.student)
        self.static = deepcopy(self.student)
        self.trainable = [k for k, v in self.student.items() if _is_trainable(k, v)]
        self.proto = tuple(t.clone() for t in proto_state)
        self.tau = self.cfg["TAU"]
        self.stats = monitor.WindowStats(self.cfg["AVG_MONITOR_SIZE"], self.cfg["EXP_MONITOR_CONST"], "hamming")
        self.switch = monitor.SwitchState(self.cfg["GRAY_AREA"], self.cfg["DEV_THRESH"])
        self.momentum = {}
        self.first_step = first_step
        self.g0, self.g1 = optim.param_groups(list(self.student.keys()))

    # ---- pieces -----------------------------------------------------------------
    def _student_forward(self, image, track, mask):
        for k in self.trainable:
            self.student[k].requires_grad_(True)
        return model.forward(image, self.student, model.BNMode(True, track, 0.1), mask)[1]

    def _backward(self, loss, grads):
        names = [k for k in self.trainable if not k.startswith("layer5.")]
        gs = torch.autograd.grad(loss, [self.student[k] for k in names])
        for k, g in zip(names, gs):
            grads[k] = g if k not in grads else grads[k] + g

    def _dynamic_prior(self, image):
        dy = model.forward(image, self.dynamic, model.BNMode(False))[1]
        p_dy = dy["out"].softmax(1)
        self.stats.add({"prior dynamic": p_dy.max(1)[0].mean().item()})
        return p_dy

    @torch.no_grad()
    def teacher_labels(self, image, mask):
        c = self.cfg
        ema = model.forward(image, self.ema, model.BNMode(True, True, 0.1), mask)[1]
        prior_ema = ema["out"].softmax(1)
        self.stats.add({"prior EMA": prior_ema.max(1)[0].mean().item()})
        prior = c["EMA_LAMBDA"] * prior_ema
        if c["STATIC_LAMBDA"] > 0:
            st = model.forward(image, self.static, model.BNMode(False))[1]
            p_st = st["out"].softmax(1)
            self.stats.add({"prior static": p_st.max(1)[0].mean().item()})
            prior = prior + c["STATIC_LAMBDA"] * p_st
        thr = c.get("SWITCH_PRIOR_THRESH", 0)
        if self.method == "hybrid":
            self.switch.evaluate(self.stats.avg("prior static"), self.stats.dev_avg("prior static"))
            if self.switch.current == self.switch.DYNAMIC and c["DYNAMIC_LAMBDA"] > 0:
                prior = c["DYNAMIC_LAMBDA"] * self._dynamic_prior(image)
        elif self.method == "online":
            calculate, replace = True, False
            if thr > 0 and self.stats.avg("prior static") < thr:
                replace = True
            elif thr > 0:
                calculate = False
            if c["DYNAMIC_LAMBDA"] > 0 and calculate:
                p_dy = c["DYNAMIC_LAMBDA"] * self._dynamic_prior(image)
                prior = p_dy if replace else prior + p_dy
        elif self.method == "hswitch":
            if c.get("SOFT_TRANS", True):
                share = max(min(self.stats.avg("prior static") * (25.0 / 3) - (41.0 / 6), 1), 0)
            else:
                share = int(self.stats.avg("prior static") > thr)
            self.stats.add({"percentage_static": share})
            prior = prior * share
            if c["DYNAMIC_LAMBDA"] > 0 and share < 1:
                prior = prior + (1 - share) * c["DYNAMIC_LAMBDA"] * self._dynamic_prior(image)
        elif self.method == "vswitch":
            dev = self.stats.dev_avg("prior static")
            if dev > thr:
                self.switch.current = self.switch.STATIC
            elif dev < -thr:
                self.switch.current = self.switch.DYNAMIC
            if self.switch.current == self.switch.DYNAMIC and c["DYNAMIC_LAMBDA"] > 0:
                prior = c["DYNAMIC_LAMBDA"] * self._dynamic_prior(image)
        else:
            raise ValueError(self.method)
        self.stats.add({"prior": prior.max(1)[0].mean().item()})
        labels, soft, conf = prototypes.assign(ema["feat"], prior, self.proto, self.tau,
                                               c["PSEUDO_THRESH"], c["DISTANCE_MEASURE"])
        self.stats.add({"prototypes": conf.item()})
        if self.stats.avg("prototypes") > c["CONF_REG_THRESH"]:
            self.tau += 0.001
            self.stats.add({"tau": self.tau})
        self.stats.add({"pseudolabel confidence": soft.max(1)[0].mean().item()})
        return ema, labels, soft

    @torch.no_grad()
    def initial_prototypes(self, source_batches):
        """calculate_prototypes with STARTING_PROTO "source" (prototypes.py:128-155, called
        under switch_batch_statistics(False) at :473-478): the student runs in train mode
        (batch-stat BN, Dropout2d active, one mask draw per batch), classes come from the
        nearest-resized ground truth, ignored pixels are dropped."""
        state = None
        for b in source_batches:
            mask = model.draw_drop_mask(b["image"].shape[0])
            pred = model.forward(b["image"], self.student, model.BNMode(True, False, 0.1), mask)[1]
            k, h, w = pred["out"].shape[1:]
            lab = torch.nn.functional.interpolate(b["label"].unsqueeze(1).float(), size=(h, w)).view(-1)
            keep = lab != 255
            rows = pred["feat"].permute(1, 0, 2, 3).reshape(pred["feat"].shape[1], -1)[:, keep].T
            onehot = torch.nn.functional.one_hot(lab[keep].long(), k)
            state = prototypes.running_append(state, rows, onehot)
        return state

    # ---- the step ---------------------------------------------------------------
    def step(self, batch_src, batch_trg, masks=(None, None, None)):
        c = self.cfg
        grads, log = {}, {}
        # source replay batch: batch-stat BN, running stats frozen (BN_POLICY freeze)
        out_s = self._student_forward(batch_src["image"], False, masks[0])["out"]
        ce_s = losses.ce_hard(out_s, batch_src["label_res"])
        src_total = c["BUFF_CE"] * ce_s
        self._backward(src_total, grads)
        src_log = {"buff_ce_loss": ce_s.detach(), "buff_rce_loss": 0, "buff_loss": src_total.detach()}
        # target batch
        pred = self._student_forward(batch_trg["image"], True, masks[1])
        out_t = pred["out"]
        self.stats.add({"model": out_t.detach().softmax(1).max(1)[0].mean().item()})
        ema, labels, soft = self.teacher_labels(batch_trg["image"], masks[2])
        self.proto = prototypes.ema_update(self.proto, ema["feat"], ema["out"], c["MA_LAMBDA"])
        b, k, h, w = out_t.shape
        pseudo = labels.reshape(b, h, w)
        parts = losses.target_loss(out_t, pseudo, c["RCE_ALPHA"], c["RCE_BETA"], c["REGULARIZER_WEIGHT"])
        self._backward(parts["Total target loss"], grads)
        log.update({n: v.detach() for n, v in parts.items()})
        log["pseudolabel_pixel_num"] = ((labels >= 0) & (labels != 255)).float().sum()
        log["output & prototype agreement"] = (pseudo == out_t.argmax(1)).float().mean()
        log["mean_prototype_intensity_values"] = (self.proto[0] ** 2).mean()
        log["JS Divergance loss"] = 0  # JS_D: 0 and no EWC in hybrid_switch.yml
        log["model regularization"] = 0
        for n, v in self.stats.avg().items():
            log[f"{n} confidence ma"] = v
        for n, v in self.stats.exp().items():
            log[f"{n} exp confidence ma"] = v
        log["dev avg prior static"] = self.stats.dev_avg("prior static")
        log["encoder_lr"] = c["LEARNING_RATE"] * c["LR_RATIO"][0]
        log.update(src_log)
        self.last = {"soft": soft, "labels": labels, "ema": ema, "out": out_t.detach(), "grads": grads}
        # optimizer step (poly schedule with POWER 0 -> constant rates)
        with torch.no_grad():
            for k_, t in self.student.items():
                t.requires_grad_(False)
            lr0, lr1 = (c["LEARNING_RATE"] * r for r in c["LR_RATIO"])
            for name, times in self.g0:
                if name in grads:
                    self.momentum[name] = optim.sgd_apply(self.student[name], grads[name], self.momentum.get(name),
                                                          lr0, times, c["MOMENTUM"], c["WEIGHT_DECAY"], self.first_step)
            for name in self.g1:
                if name in grads:
                    self.momentum[name] = optim.sgd_apply(self.student[name], grads[name], self.momentum.get(name),
                                                          lr1, 1, c["MOMENTUM"], c["WEIGHT_DECAY"], self.first_step)
        return log

    def step_sharded(self, shards, masks=None):
        """The step of N data-parallel ranks, emulated sequentially: `shards` = [(batch_src, batch_trg)] one
        per rank, `masks` = [(source, target-student, teacher) Dropout2d masks] per rank.  Every rank normalises with
        the batch statistics of its own micro-batch and starts from the same running statistics (averaged afterwards);
        gradients are averaged; the teacher/static confidences are averaged before the ONE switch decision; pseudo-labels
        of every rank use the pre-step prototypes; the class statistics are summed before the prototype EMA; the
        monitor sees rank-means.  Hybrid method only.  Returns rank 0's log dict."""
        assert self.method == "hybrid"
        c, n = self.cfg, len(shards)
        masks = masks or [(None, None, None)] * n
        grads, log = {}, {}
        for r, ((src, _), m) in enumerate(zip(shards, masks)):
            ce = losses.ce_hard(self._student_forward(src["image"], False, m[0])["out"], src["label_res"])
            self._backward(c["BUFF_CE"] * ce / n, grads)
            if r == 0:
                log.update({"buff_ce_loss": ce.detach(), "buff_rce_loss": 0, "buff_loss": (c["BUFF_CE"] * ce).detach()})
        teachers = []
        with torch.no_grad():
            for (_, trg), m in zip(shards, masks):
                ema = model.forward(trg["image"], self.ema, model.BNMode(True, True, 0.1), m[2])[1]
                st = model.forward(trg["image"], self.static, model.BNMode(False))[1]
                teachers.append((ema, ema["out"].softmax(1), st["out"].softmax(1)))
        self.stats.add({"prior EMA": sum(t[1].max(1)[0].mean().item() for t in teachers) / n})
        self.stats.add({"prior static": sum(t[2].max(1)[0].mean().item() for t in teachers) / n})
        self.switch.evaluate(self.stats.avg("prior static"), self.stats.dev_avg("prior static"))
        use_dynamic = self.switch.current == self.switch.DYNAMIC and c["DYNAMIC_LAMBDA"] > 0
        running = [k for k in self.student if "running_" in k]
        start = {k: self.student[k].clone() for k in running}
        tracked = {k: self.student[k].clone() for k in self.student if k.endswith("num_batches_tracked")}
        mean_running = {k: torch.zeros_like(v) for k, v in start.items()}
        seen = {key: 0.0 for key in ("model", "prior dynamic", "prior", "prototypes", "pseudolabel confidence")}
        sums = None
        for r, ((_, trg), m, (ema, p_ema, p_st)) in enumerate(zip(shards, masks, teachers)):
            for k in running:
                self.student[k] = start[k].clone()
            for k, v in tracked.items():
                self.student[k] = v.clone()
            out_t = self._student_forward(trg["image"], True, m[1])["out"]
            for k in running:
                mean_running[k] += self.student[k].detach() / n
            seen["model"] += out_t.detach().softmax(1).max(1)[0].mean().item() / n
            with torch.no_grad():
                prior = c["EMA_LAMBDA"] * p_ema + c["STATIC_LAMBDA"] * p_st
                if use_dynamic:
                    p_dy = model.forward(trg["image"], self.dynamic, model.BNMode(False))[1]["out"].softmax(1)
                    seen["prior dynamic"] += p_dy.max(1)[0].mean().item() / n
                    prior = c["DYNAMIC_LAMBDA"] * p_dy
                seen["prior"] += prior.max(1)[0].mean().item() / n
                labels, soft, conf = prototypes.assign(ema["feat"], prior, self.proto, self.tau, c["PSEUDO_THRESH"],
                                                       c["DISTANCE_MEASURE"])
                seen["prototypes"] += conf.item() / n
                seen["pseudolabel confidence"] += soft.max(1)[0].mean().item() / n
                s1, cnt = prototypes.class_sums(ema["feat"], ema["out"])
                s2, _ = prototypes.class_sums(ema["feat"] ** 2, ema["out"])
                sums = [s1, s2, cnt] if sums is None else [a + b for a, b in zip(sums, (s1, s2, cnt))]
            b, k, h, w = out_t.shape
            pseudo = labels.reshape(b, h, w)
            parts = losses.target_loss(out_t, pseudo, c["RCE_ALPHA"], c["RCE_BETA"], c["REGULARIZER_WEIGHT"])
            self._backward(parts["Total target loss"] / n, grads)
            if r == 0:
                log.update({name: v.detach() for name, v in parts.items()})
                log["pseudolabel_pixel_num"] = ((labels >= 0) & (labels != 255)).float().sum()
                log["output & prototype agreement"] = (pseudo == out_t.argmax(1)).float().mean()
                self.last = {"soft": soft, "labels": labels}
        with torch.no_grad():
            for k_, t in self.student.items():
                t.requires_grad_(False)
            for k in running:
                self.student[k] = mean_running[k]
            # prototype EMA from the summed class statistics (prototype_handler.ma, :88-99)
            proto, sqmean, counter = self.proto
            s1, s2, cnt = sums
            keep = c["MA_LAMBDA"] ** (cnt > 0).float()
            safe = torch.where(cnt > 0, cnt, torch.ones_like(cnt))
            self.proto = ((proto.T * keep).T + ((1 - keep) * (s1.T / safe)).T, (sqmean.T * keep).T + ((1 - keep) * (s2.T / safe)).T,
                          counter)
            for key, value in seen.items():
                if key != "prior dynamic" or use_dynamic:
                    self.stats.add({key: value})
            log["mean_prototype_intensity_values"] = (self.proto[0] ** 2).mean()
            for name, v in self.stats.avg().items():
                log[f"{name} confidence ma"] = v
            lr0, lr1 = (c["LEARNING_RATE"] * r_ for r_ in c["LR_RATIO"])
            for name, times in self.g0:
                if name in grads:
                    self.momentum[name] = optim.sgd_apply(self.student[name], grads[name], self.momentum.get(name), lr0, times,
                                                          c["MOMENTUM"], c["WEIGHT_DECAY"], self.first_step)
            for name in self.g1:
                if name in grads:
                    self.momentum[name] = optim.sgd_apply(self.student[name], grads[name], self.momentum.get(name), lr1, 1,
                                                          c["MOMENTUM"], c["WEIGHT_DECAY"], self.first_step)
        return log

    @torch.no_grad()
    def update_ema(self):
        a = self.cfg["EMA_UPDATE"]
        for k_, q in self.student.items():
            if "running_" in k_ or k_.endswith("num_batches_tracked"):
                self.ema[k_] = q.clone()
            else:
                self.ema[k_] = self.ema[k_] * a + q * (1.0 - a)

    def refresh_dynamic(self):
        """update_dynamic (prototypes.py:99-102): dynamic := copy of the student."""
        self.dynamic = {k: v.detach().clone() for k, v in self.student.items()}
