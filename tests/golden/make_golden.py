#!/usr/bin/env python3
"""Generate the golden vectors (G1-G15) by IMPORTING the reference on CPU.

Run in the build container only (the reference does not exist on the GPU box):

    python tests/golden/make_golden.py            # writes tests/golden/*.npz

Only inputs' seeds and the reference's OUTPUTS are stored; weights and batches are
re-created on any machine by ``onda_amd.synthetic``.  Three shims let the reference
import here: stub ``wandb`` / ``addict`` modules (tests/golden/_stubs, own code) and a
``yaml.load`` wrapper that supplies the Loader PyYAML >= 6 requires.
"""
import json
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("ONDA_REFERENCE", "/root/reference")
sys.path[:0] = [REF, os.path.join(HERE, "_stubs"), ROOT]

import numpy as np  # noqa: E402
import torch  # noqa: E402
import yaml  # noqa: E402

_yaml_load = yaml.load
yaml.load = lambda f, Loader=None: _yaml_load(f, Loader=Loader or yaml.SafeLoader)

from addict import Dict  # noqa: E402  (stub)
from framework.model.deeplabv2 import get_deeplab_v2  # noqa: E402
from framework.domain_adaptation.methods.prototype_handler import prototype_handler  # noqa: E402
from framework.domain_adaptation.methods.prototypes import regular_loss  # noqa: E402
from framework.domain_adaptation.methods.prototypes_hybrid_switch import hybrid_proDA, model_select  # noqa: E402
from framework.domain_adaptation.methods.adaptation_model import switch_batch_statistics  # noqa: E402
from framework.utils.func import loss_calc  # noqa: E402
from framework.utils.loss import rce  # noqa: E402
from framework.utils.monitoring import Monitor  # noqa: E402

from onda_amd.synthetic import fill_state_dict, synth_batch, synth_prototypes  # noqa: E402

torch.set_num_threads(8)


def save(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = v
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, {k: getattr(v, "shape", None) for k, v in out.items()})


def ref_model(seed, head_scale):
    m = get_deeplab_v2(num_classes=19, multi_level=True, layers=[3, 4, 6, 3], classifier="ProDA")
    m.multi_level = False  # model_handler.py:58 with MODEL.MULTI_LEVEL False
    fill_state_dict(m, seed, head_scale)
    return m


def digest(t, n=256):
    """(sum, abs-sum, strided sample) of a tensor: pins big tensors in little space."""
    f = t.detach().double().reshape(-1)
    idx = (torch.arange(n, dtype=torch.int64) * f.numel()) // n
    return np.concatenate([[f.sum().item(), f.abs().sum().item()], f[idx].numpy()])


def interp_argmax(out, size):
    up = torch.nn.Upsample(size=size, mode="bilinear", align_corners=True)(out)  # adaptation_model.py:94-98
    return up, up.softmax(axis=1).argmax(dim=1)  # :145,153


# ------------------------------------------------------------------------------------- G1
def g1():
    m = ref_model(1, 3.0).eval()
    b = synth_batch(2, 64, 128, seed=7)
    with torch.no_grad():
        _, o = m(b["image"])
        up, amap = interp_argmax(o["out"], (64, 128))
    save("g1_eval_small", feat=o["feat"], out=o["out"], up=up, argmax=amap.to(torch.uint8))
    b = synth_batch(1, 512, 1024, seed=8)
    with torch.no_grad():
        _, o = m(b["image"])
        up, amap = interp_argmax(o["out"], (512, 1024))
    top2 = up.topk(2, dim=1)[0]
    margin = (top2[:, 0] - top2[:, 1])
    save("g1_eval_large", out=o["out"], feat_digest=digest(o["feat"], 4096), argmax=amap.to(torch.uint8),
         margin_f16=margin.to(torch.float16))


# ------------------------------------------------------------------------------------- G2
def g2():
    b = synth_batch(2, 64, 128, seed=7)
    res = {}
    for track in (True, False):
        m = ref_model(1, 3.0).train()
        switch_batch_statistics(m, track)
        torch.manual_seed(2024)
        _, o = m(b["image"])
        loss = loss_calc(o["out"], b["label_res"], "cpu")
        loss.backward()
        tag = "track" if track else "frozen"
        res[f"{tag}_out"], res[f"{tag}_feat"], res[f"{tag}_loss"] = o["out"], o["feat"], loss
        sd = m.state_dict()
        for k in ("bn1", "layer1.0.bn1", "layer2.0.downsample.1", "layer3.2.bn2", "layer4.2.bn3"):
            res[f"{tag}_{k}.running_mean"] = sd[k + ".running_mean"]
            res[f"{tag}_{k}.running_var"] = sd[k + ".running_var"]
            res[f"{tag}_{k}.num_batches_tracked"] = sd[k + ".num_batches_tracked"]
        if track:
            names, dig = [], []
            for n, p in m.named_parameters():
                if p.grad is not None:
                    names.append(n)
                    dig.append(digest(p.grad))
            res["grad_names"] = np.array(names)
            res["grad_digest"] = np.stack(dig)
            res["grad_layer6.head.1.weight"] = dict(m.named_parameters())["layer6.head.1.weight"].grad
            res["grad_layer6.bottleneck.2.weight"] = dict(m.named_parameters())["layer6.bottleneck.2.weight"].grad
            res["grad_layer6.conv2d_list.0.0.bias"] = dict(m.named_parameters())["layer6.conv2d_list.0.0.bias"].grad
            res["grad_conv1.weight"] = m.conv1.weight.grad
    torch.manual_seed(2024)
    mask = torch.empty(2, 256, 1, 1).bernoulli_(0.9).div_(0.9)
    res["drop_mask"] = mask
    save("g2_train_small", **res)


# ------------------------------------------------------------------------------------- G3
def g3():
    g = torch.Generator().manual_seed(31)
    res = {}
    for case, frac in (("mixed", 0.3), ("none_ignored", 0.0), ("all_ignored", 1.0)):
        logits = (3 * torch.randn(2, 19, 9, 17, generator=g)).requires_grad_(True)
        target = torch.randint(0, 19, (2, 9, 17), generator=g)
        target[torch.rand(2, 9, 17, generator=g) < frac] = 255
        ce = loss_calc(logits, target, "cpu")
        r = rce(logits, target, "cpu")
        reg = regular_loss("MRKLD", logits)
        res[f"{case}_logits"], res[f"{case}_target"] = logits, target
        res[f"{case}_ce"], res[f"{case}_rce"], res[f"{case}_mrkld"] = ce, r, reg
        # (all_ignored: ce is NaN = the mean over an empty selection, but its GRADIENT is exact zeros -- autograd scatters
        #  an empty gradient back through predict[mask] -- so the total's gradient is finite: RCE + MRKLD only)
        total = 0.1 * ce + 1.0 * r + 0.1 * reg
        res[f"{case}_grad"] = torch.autograd.grad(total, logits)[0]
        res[f"{case}_grad_ce"] = torch.autograd.grad(loss_calc(logits, target, "cpu"), logits)[0]
    save("g3_losses", **res)


# ------------------------------------------------------------------------------------- G4
def g4():
    g = torch.Generator().manual_seed(41)
    res = {}
    proto, sq, cnt = synth_prototypes()
    # regime "far": features sit near one prototype each; regime "near": prototypes close
    # together so that the prior and the threshold decide
    base = torch.randn(1, 256, generator=g)
    proto_near = base + 0.15 * torch.randn(19, 256, generator=g)
    sq_near = proto_near ** 2 + (0.5 + torch.rand(19, 256, generator=g)) ** 2
    for regime, (P, S) in (("far", (proto, sq)), ("near", (proto_near, sq_near))):
        cls = torch.randint(0, 19, (2 * 9 * 17,), generator=g)
        feat_rows = P[cls] + 0.7 * torch.randn(cls.numel(), 256, generator=g)
        feat = feat_rows.reshape(2, 9, 17, 256).permute(0, 3, 1, 2).contiguous()
        prior = (2 * torch.randn(2, 19, 9, 17, generator=g)).softmax(1)
        out = torch.randn(2, 19, 9, 17, generator=g)
        res[f"{regime}_proto"], res[f"{regime}_sqmean"], res[f"{regime}_counter"] = P, S, cnt
        res[f"{regime}_feat"], res[f"{regime}_prior"], res[f"{regime}_out"] = feat, prior, out
        for metric in ("mahalanobis", "euclidean"):
            for tau in (1, 2):
                for thresh in (0, 0.3):
                    h = prototype_handler(0.9995, tau, thresh, metric)
                    h.prototypes, h.squared_mean, h.counter = P.clone(), S.clone(), cnt.clone()
                    tag = f"{regime}_{metric}_t{tau}_th{thresh}"
                    res[tag + "_labels"] = h.pseudo_labels(feat, prior)
                    res[tag + "_soft"] = h.pseudo_labels(feat, prior, soft=True)
        h = prototype_handler(0.9995, 1, 0.3, "mahalanobis")
        h.prototypes, h.squared_mean, h.counter = P.clone(), S.clone(), cnt.clone()
        res[f"{regime}_global_var"] = h.global_var()
        h.ma(feat, out)
        res[f"{regime}_ma_proto"], res[f"{regime}_ma_sqmean"] = h.prototypes, h.squared_mean
        h2 = prototype_handler(0.9995, 1, 0.3, "mahalanobis")
        h2.append(feat, out)
        h2.append(feat * 0.5 + 0.1, out.flip(0))
        res[f"{regime}_append_proto"], res[f"{regime}_append_sqmean"], res[f"{regime}_append_counter"] = \
            h2.prototypes, h2.squared_mean, h2.counter
    save("g4_prototypes", **res)


# ------------------------------------------------------------------------------------- G5
def g5():
    rng = np.random.RandomState(5)
    n = 520
    t = np.arange(n)
    seq = 0.865 + 0.05 * np.sin(t / 60.0) + 0.004 * rng.randn(n)
    mon = Monitor(200, 0.003, "hamming")
    sel = model_select(model_select.static, [0.83, 0.9], 0.0002)
    avg, exp, dev, cur = [], [], [], []
    for v in seq:
        mon.add({"prior static": float(v)})
        a, d = mon.avg("prior static"), mon.dev_avg("prior static")
        sel.evaluate(a, d)
        avg.append(a); exp.append(mon.exp("prior static")); dev.append(d); cur.append(sel.current)
    save("g5_switch", seq=seq, avg=np.array(avg), exp=np.array(exp), dev=np.array(dev, dtype=np.float64),
         current=np.array(cur, dtype=np.int64), missing_avg=np.array(mon.avg("nope")), missing_dev=np.array(mon.dev_avg("nope")))


# ------------------------------------------------------------------------------------- G6
def g6():
    g = torch.Generator().manual_seed(61)
    w3 = torch.nn.Parameter(torch.randn(5, 4, generator=g))
    w4 = torch.nn.Parameter(torch.randn(7, generator=g))
    w1 = torch.nn.Parameter(torch.randn(3, 3, generator=g))
    h1 = torch.nn.Parameter(torch.randn(6, generator=g))
    init = [p.detach().clone() for p in (w3, w4, w1, h1)]
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        opt = torch.optim.SGD([{"params": [w3, w4, w1, w3, w4, w3, w4, w4], "lr": 8e-4},
                               {"params": [h1], "lr": 1e-4}], lr=1e-5, momentum=0.9, weight_decay=1e-4, foreach=False)
    grads, traj = [], []
    for _ in range(3):
        gs = [torch.randn(p.shape, generator=g) for p in (w3, w4, w1, h1)]
        for p, gr in zip((w3, w4, w1, h1), gs):
            p.grad = gr.clone()
        opt.step()
        grads.append(gs)
        traj.append([p.detach().clone() for p in (w3, w4, w1, h1)])
    res = {}
    for i, n in enumerate(("w3", "w4", "w1", "h1")):
        res[n + "_init"] = init[i]
        for s in range(3):
            res[f"{n}_grad{s}"], res[f"{n}_after{s}"] = grads[s][i], traj[s][i]
    # multiplicity pattern of the reference's own parameter groups
    m = ref_model(1, 1.0)
    name_of = {id(p): n for n, p in m.named_parameters()}
    groups = m.optim_parameters(1.0)
    c0 = {}
    for p in groups[0]["params"]:
        c0[name_of[id(p)]] = c0.get(name_of[id(p)], 0) + 1
    g1names = [name_of[id(p)] for p in groups[1]["params"]]
    res["group0_json"] = np.array(json.dumps(c0))
    res["group1_json"] = np.array(json.dumps(g1names))
    save("g6_optimizer", **res)


# ------------------------------------------------------------------------------------- G7
def make_cfg(tmp):
    from framework.domain_adaptation import config_ouda
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        y = Dict(yaml.load(open("configs/hybrid_switch.yml")))
    finally:
        os.chdir(cwd)
    cfg = Dict()
    cfg.SCHEME.RESOLUTION = [128, 64]
    cfg.MODEL.LR_RATIO = "80:10"
    cfg.MODEL.MULTI_LEVEL = False
    cfg.TRAINING.REPLAY_BUFFER = 1000
    cfg.TRAINING.BATCH_SIZE = 2
    cfg.TRAINING.BUFFER_DYNAMIC = False
    cfg.OTHERS.DEVICE = "cpu"
    cfg.OTHERS.SNAPSHOT_DIR = tmp
    cfg.OTHERS.ECE_SKIP = True
    cfg.NUM_CLASSES = 19
    spec = y.METHOD.ADAPTATION.PROTO_ONLINE_HYBRIDSWITCH
    spec.LOAD_PROTO = None
    spec.set_ = (25,)
    return cfg, spec


def tolog(d):
    out = {}
    for k, v in d.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().double().item() if v.numel() == 1 else v.detach().numpy()
        out[k] = float(v) if np.isscalar(v) else v
    return out


def g7():
    for tag, head_scale in (("static", 40.0), ("dynamic", 3.0)):
        with tempfile.TemporaryDirectory() as tmp:
            cfg, spec = make_cfg(tmp)
            model = ref_model(1, head_scale)
            da = hybrid_proDA(model, cfg, spec)
            src = [synth_batch(2, 64, 128, seed=100 + i) for i in range(2)]
            trg = [synth_batch(2, 64, 128, seed=200 + i) for i in range(2)]
            torch.manual_seed(123)
            da.update_dynamic()
            switch_batch_statistics(da.model, False)
            da.calculate_prototypes(src)
            switch_batch_statistics(da.model, True)
            res = {"proto0": da.prototypes.prototypes.clone(), "sqmean0": da.prototypes.squared_mean.clone(),
                   "counter0": da.prototypes.counter.clone()}
            da.optimizer.zero_grad()
            import warnings
            for s in range(2):
                da.adjust_learning_rate(s, 6)
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    log = da.step([src[s]], trg[s])
                da.update_ema()
                lg = tolog(log)
                res[f"log{s}_json"] = np.array(json.dumps({k: v for k, v in lg.items() if np.isscalar(v)}))
                res[f"labels{s}"] = trg[s]["stored_predictions"].argmax(1).to(torch.uint8)
                res[f"soft{s}"] = trg[s]["stored_predictions"].to(torch.float32)
                res[f"proto{s + 1}"] = da.prototypes.prototypes.clone()
                res[f"sqmean{s + 1}"] = da.prototypes.squared_mean.clone()
                res[f"branch{s}"] = np.array(da.model_select.current)
                names, dig = [], []
                for (n, p) in da.model.state_dict().items():
                    names.append("student." + n); dig.append(digest(p.float(), 64))
                for (n, p) in da.ema_model.state_dict().items():
                    names.append("teacher." + n); dig.append(digest(p.float(), 64))
                res[f"state_names{s}"] = np.array(names)
                res[f"state_digest{s}"] = np.stack(dig)
            save(f"g7_step_{tag}", **res)
            print(tag, "branch", res["branch0"], res["branch1"], json.loads(str(res["log1_json"]))["pseudolabel_pixel_num"])


# ------------------------------------------------------------------------------------- G8
def g8():
    """The other prototype methods of the reference (same kernels, different prior mixing):
    PROTO_ONLINE with static_model.yml / dynamic_model.yml settings, PROTO_ONLINE_HSWITCH
    (confidence_switch.yml) and PROTO_ONLINE_VSWITCH (confidence_der_switch.yml); two steps each."""
    from framework.domain_adaptation.methods.prototypes import online_proDA
    from framework.domain_adaptation.methods.prototypes_hswitch import hswitch_proDA
    from framework.domain_adaptation.methods.prototypes_vswitch import vswitch_proDA
    variants = {
        "online_static": (online_proDA, dict(SWITCH_PRIOR_THRESH=1, STATIC_LAMBDA=1, DYNAMIC_LAMBDA=0), 40.0),
        "online_dynamic": (online_proDA, dict(SWITCH_PRIOR_THRESH=0, STATIC_LAMBDA=0, DYNAMIC_LAMBDA=1), 40.0),
        "hswitch": (hswitch_proDA, dict(SWITCH_PRIOR_THRESH=0.86, SOFT_TRANS=True), 9.0),
        "vswitch": (vswitch_proDA, dict(SWITCH_PRIOR_THRESH=0.0002, DEV_THRESH=0.0002), 40.0),
    }
    import warnings
    for tag, (cls, over, head_scale) in variants.items():
        with tempfile.TemporaryDirectory() as tmp:
            cfg, spec = make_cfg(tmp)
            for k in ("GRAY_AREA",):
                spec.pop(k, None)
            for k, v in over.items():
                spec[k] = v
            model = ref_model(1, head_scale)
            da = cls(model, cfg, spec)
            src = [synth_batch(2, 64, 128, seed=100 + i) for i in range(2)]
            trg = [synth_batch(2, 64, 128, seed=200 + i) for i in range(2)]
            torch.manual_seed(123)
            da.update_dynamic()
            switch_batch_statistics(da.model, False)
            da.calculate_prototypes(src)
            switch_batch_statistics(da.model, True)
            da.optimizer.zero_grad()
            res = {}
            for s in range(2):
                da.adjust_learning_rate(s, 6)
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    log = da.step([src[s]], trg[s])
                da.update_ema()
                lg = tolog(log)
                res[f"log{s}_json"] = np.array(json.dumps({k: v for k, v in lg.items() if np.isscalar(v)}))
                res[f"soft{s}"] = trg[s]["stored_predictions"].to(torch.float32)
                res[f"proto{s + 1}"] = da.prototypes.prototypes.clone()
            save(f"g8_{tag}", **res)
            print(tag, json.loads(str(res["log1_json"]))["pseudolabel_pixel_num"],
                  {k: round(v, 4) for k, v in json.loads(str(res["log0_json"])).items() if "prior" in k or "percentage" in k})


def g9():
    """Input pipeline: the reference's own `_load_img` (base_dataset.py:89-95, i.e. Pillow) on
    PNG files written here, for the resize cases of segmentation_db.py:82-96; `color_mapper`
    (func.py:88-115) for the id map.  The tensor transform is torchvision's (absent): stored as the
    formula of its published source evaluated with torch ops (unpinned, see oracle/pipeline.py)."""
    from PIL import Image
    from framework.dataset.base_dataset import _load_img
    from framework.utils.func import color_mapper
    rng = np.random.default_rng(9)
    res = {}
    cases = [(64, 96, 48, 32), (50, 37, 23, 31), (48, 64, 100, 70), (128, 256, 128, 64), (77, 33, 77, 66)]
    # the label table: Cityscapes ids 0..33 -> 19 train ids, 255 elsewhere (dataset json of the reference run)
    ids = {i: 255 for i in range(34)}
    for tid, i in enumerate([7, 8, 11, 12, 13, 17, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 31, 32, 33]):
        ids[i] = tid
    if not hasattr(np, "int"):
        np.int = int  # func.py:107 uses the alias numpy removed
    mapper = color_mapper(ids)
    res["lut"] = np.asarray(mapper.color_map, np.int32)
    mean, std = np.array([123.675, 116.28, 103.53]), np.array([58.395, 57.12, 57.375])  # hybrid_switch.yml:9-10
    res["mean"], res["std"] = mean, std
    with tempfile.TemporaryDirectory() as tmp:
        for n, (h, w, W, H) in enumerate(cases):
            # smooth + noise, so that the antialiased filter sees structure as well as extremes
            yy, xx = np.mgrid[0:h, 0:w]
            base = (127 + 120 * np.sin(xx / 7.0 + n)[..., None] * np.cos(yy / 5.0)[..., None] * np.array([1, .5, -1])).clip(0, 255)
            img = (base + rng.integers(-40, 41, (h, w, 3))).clip(0, 255).astype(np.uint8)
            img[:3, :5] = 255
            img[-4:, -2:] = 0
            lab = rng.integers(0, 34, (h, w), dtype=np.uint8)
            fi, fl = os.path.join(tmp, f"i{n}.png"), os.path.join(tmp, f"l{n}.png")
            Image.fromarray(img).save(fi)
            Image.fromarray(lab).save(fl)
            out = _load_img(fi, (W, H), Image.BICUBIC, rgb=True)
            res[f"img{n}"], res[f"size{n}"], res[f"resized{n}"] = img, np.array([W, H]), out
            t = torch.from_numpy(out[:, :, ::-1].copy().transpose(2, 0, 1)).contiguous().to(torch.float32).div(255)
            m32, s32 = torch.as_tensor(mean / 255, dtype=torch.float32), torch.as_tensor(std / 255, dtype=torch.float32)
            res[f"tensor{n}"] = t.sub_(m32[:, None, None]).div_(s32[:, None, None])
            res[f"lab{n}"] = lab
            res[f"label{n}"] = mapper(_load_img(fl, (W, H), Image.NEAREST, rgb=False)).astype(np.uint8)
            res[f"label_res{n}"] = mapper(_load_img(fl, [int(x / 8 + 1) for x in (W, H)], Image.NEAREST, rgb=False)).astype(np.uint8)
    res["ncases"] = np.array(len(cases))
    save("g9_pipeline", **res)


def g10():
    """One full hybrid_proDA step (+update_ema) at the BASELINE size -- 512x1024, batch 4, static branch -- on the
    reference (CPU: a few minutes, ~12 GB): log dict, pseudo-label map with its tie mask, prototypes before / after,
    strided digests of every student / teacher tensor after the step."""
    import warnings
    with tempfile.TemporaryDirectory() as tmp:
        cfg, spec = make_cfg(tmp)
        cfg.SCHEME.RESOLUTION = [1024, 512]
        cfg.TRAINING.BATCH_SIZE = 4
        model = ref_model(1, 40.0)
        da = hybrid_proDA(model, cfg, spec)
        src = [synth_batch(4, 512, 1024, seed=1000 + i) for i in range(2)]
        trg = synth_batch(4, 512, 1024, seed=2000)
        torch.manual_seed(123)
        da.update_dynamic()
        switch_batch_statistics(da.model, False)
        da.calculate_prototypes(src)
        switch_batch_statistics(da.model, True)
        res = {"proto0": da.prototypes.prototypes.clone(), "counter0": da.prototypes.counter.clone()}
        da.optimizer.zero_grad()
        da.adjust_learning_rate(0, 6)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            log = da.step([src[0]], trg)
        da.update_ema()
        lg = tolog(log)
        res["log_json"] = np.array(json.dumps({k: v for k, v in lg.items() if np.isscalar(v)}))
        soft = trg["stored_predictions"].to(torch.float32)  # [4,19,65,129]
        top2 = soft.topk(2, dim=1)[0]
        res["labels"] = soft.argmax(1).to(torch.uint8)
        res["tie_mask"] = np.packbits(((top2[:, 0] - top2[:, 1]) < 2e-3).numpy())
        res["soft_digest"] = digest(soft, 4096)
        res["soft_max"] = soft.max(1)[0].to(torch.float16)
        res["proto1"], res["sqmean1"] = da.prototypes.prototypes.clone(), da.prototypes.squared_mean.clone()
        res["branch"] = np.array(da.model_select.current)
        names, dig = [], []
        for who, mod in (("student.", da.model), ("teacher.", da.ema_model)):
            for n, p in mod.state_dict().items():
                names.append(who + n)
                dig.append(digest(p.float(), 64))
        res["state_names"], res["state_digest"] = np.array(names), np.stack(dig)
        save("g10_step_full", **res)
        print("g10 branch", res["branch"], {k: round(v, 5) for k, v in lg.items() if np.isscalar(v)})


def _full_step(name, width, height, batch, head_scale, seeds=(1000, 2000)):
    """One hybrid_proDA step (+update_ema) of the reference at a full size: the recipe of g10 with the size, the batch and
    the head scale (= which side of the switch the synthetic state sits on) as arguments."""
    import warnings
    with tempfile.TemporaryDirectory() as tmp:
        cfg, spec = make_cfg(tmp)
        cfg.SCHEME.RESOLUTION = [width, height]
        cfg.TRAINING.BATCH_SIZE = batch
        model = ref_model(1, head_scale)
        da = hybrid_proDA(model, cfg, spec)
        src = [synth_batch(batch, height, width, seed=seeds[0] + i) for i in range(2)]
        trg = synth_batch(batch, height, width, seed=seeds[1])
        torch.manual_seed(123)
        da.update_dynamic()
        switch_batch_statistics(da.model, False)
        da.calculate_prototypes(src)
        switch_batch_statistics(da.model, True)
        res = {"proto0": da.prototypes.prototypes.clone(), "counter0": da.prototypes.counter.clone()}
        da.optimizer.zero_grad()
        da.adjust_learning_rate(0, 6)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            log = da.step([src[0]], trg)
        da.update_ema()
        lg = tolog(log)
        res["log_json"] = np.array(json.dumps({k: v for k, v in lg.items() if np.isscalar(v)}))
        soft = trg["stored_predictions"].to(torch.float32)
        top2 = soft.topk(2, dim=1)[0]
        res["labels"] = soft.argmax(1).to(torch.uint8)
        res["tie_mask"] = np.packbits(((top2[:, 0] - top2[:, 1]) < 2e-3).numpy())
        res["soft_digest"] = digest(soft, 4096)
        res["soft_max"] = soft.max(1)[0].to(torch.float16)
        res["proto1"], res["sqmean1"] = da.prototypes.prototypes.clone(), da.prototypes.squared_mean.clone()
        res["branch"] = np.array(da.model_select.current)
        names, dig = [], []
        for who, mod in (("student.", da.model), ("teacher.", da.ema_model)):
            for n, p in mod.state_dict().items():
                names.append(who + n)
                dig.append(digest(p.float(), 64))
        res["state_names"], res["state_digest"] = np.array(names), np.stack(dig)
        res["shape"] = np.array([batch, height, width])
        save(name, **res)
        print(name, "branch", res["branch"], {k: round(v, 5) for k, v in lg.items() if np.isscalar(v)})


def g12():
    """The step bench.py TIMES (BASELINE config 3 as the default line runs it): 512x1024, batch 4, head x1 = the DYNAMIC
    branch, source seeds 1000/1001, target seed 2000."""
    _full_step("g12_step_bench_dynamic", 1024, 512, 4, 1.0)


def g13():
    """One adaptation step at BASELINE config 5's resolution, 1024x2048 (feature grid 129x257).  Batch 2, not 4: the
    reference's CPU step holds ~12 GB per 512x1024 bs-4 step, i.e. ~48 GB at 1024x2048 bs 4 -- more than this 64 GB
    container can give it beside the build; the batch changes the pixel count M only, not a single kernel shape."""
    _full_step("g13_step_1024x2048", 2048, 1024, 2, 40.0, seeds=(1300, 2300))


def g13b():
    """The same at batch 4 -- BASELINE config 5's own batch, the one bench.py's config-5 line runs: ~45 GB and ~15 minutes on the
    CPU of the 64 GB build container with nothing else resident (run under `ulimit -v` so that a shortfall raises instead
    of taking the container down)."""
    _full_step("g13_step_1024x2048_bs4", 2048, 1024, 4, 40.0, seeds=(1300, 2300))


def g11():
    """Eval-mode forward of one full-resolution frame of BASELINE config 5 (1 x 3 x 1024 x 2048): class map of
    interp(out).softmax.argmax, its tie mask, the logits on a 4-pixel grid and digests."""
    m = ref_model(1, 3.0).eval()
    b = synth_batch(1, 1024, 2048, seed=11)
    with torch.no_grad():
        _, o = m(b["image"])
        up, amap = interp_argmax(o["out"], (1024, 2048))
    top2 = up.topk(2, dim=1)[0]
    tie = (top2[:, 0] - top2[:, 1]) < 1e-3 * up.abs().max()
    save("g11_eval_1024x2048", argmax=amap.to(torch.uint8), tie_mask=np.packbits(tie.numpy()), out_grid=o["out"][:, :, ::4, ::4],
         out_digest=digest(o["out"], 4096), feat_digest=digest(o["feat"], 4096), out_absmax=o["out"].abs().max())


# ------------------------------------------------------------------------------------- G14
G14 = dict(head_scale=7.0, monitor=4, perc_fill=0.0009, np_seed=3, steps=3, val_frames=10, lr=2e-7)


# ------------------------------------------------------------------------------------- G15
G15 = {"warm_steps": 3, "warm_lr_div": 50.0, "threads": (8, 3)}


def g15():
    """ONE step at the yml's own learning rate from a WARM state -- three steps at a rate 50x lower (the rate G14 pins a whole
    trajectory at) -- captured TWICE, with 8 and with 3 CPU threads: the distance between the two captures is the reference's
    own noise floor for that step (torch's CPU convolutions sum in another order), and the HIP step is held to a small
    multiple of it (tests/test_hip_model.py::test_step_at_the_yml_learning_rate_from_a_warm_state).  G7 pins a cold start
    (step 0: 2 %, step 1: 60 %, both at the yml's rate on random weights); this brackets the step the trajectory tests
    leave open -- the yml's rate on a state that several optimizer steps have already moved."""
    import warnings
    runs = {}
    for threads in G15["threads"]:
        torch.set_num_threads(threads)
        with tempfile.TemporaryDirectory() as tmp:
            cfg, spec = make_cfg(tmp)
            lr = float(spec.LEARNING_RATE)
            model = ref_model(1, 40.0)
            da = hybrid_proDA(model, cfg, spec)
            n = G15["warm_steps"] + 1
            src = [synth_batch(2, 64, 128, seed=500 + i) for i in range(n)]
            trg = [synth_batch(2, 64, 128, seed=600 + i) for i in range(n)]
            torch.manual_seed(123)
            da.update_dynamic()
            switch_batch_statistics(da.model, False)
            da.calculate_prototypes(src[:2])
            switch_batch_statistics(da.model, True)
            da.optimizer.zero_grad()
            logs, states = [], []
            for s_ in range(n):
                da.cfg_spec.LEARNING_RATE = lr / G15["warm_lr_div"] if s_ < G15["warm_steps"] else lr
                da.adjust_learning_rate(s_, 8)
                if s_ == n - 1:
                    states.append({k: v.detach().double().clone() for k, v in da.model.state_dict().items()})
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    log = da.step([src[s_]], trg[s_])
                da.update_ema()
                logs.append({k: v for k, v in tolog(log).items() if np.isscalar(v)})
            states.append({k: v.detach().double().clone() for k, v in da.model.state_dict().items()})
            runs[threads] = (logs, states, int(da.model_select.current))
    torch.set_num_threads(8)
    (logs_a, st_a, br_a), (logs_b, st_b, br_b) = runs[G15["threads"][0]], runs[G15["threads"][1]]
    names = [k for k, v in st_a[0].items() if v.is_floating_point() and v.dim() > 0]
    num = den = 0.0
    upd_a, upd_b, pre = [], [], []
    for k in names:
        ua, ub = st_a[1][k] - st_a[0][k], st_b[1][k] - st_b[0][k]
        num += float(((ua - ub) ** 2).sum())
        den += float((ua ** 2).sum())
        upd_a.append(digest(ua, 64))
        upd_b.append(digest(ub, 64))
        pre.append(digest(st_a[0][k], 64))
    floor = (num / den) ** 0.5
    print("g15: the reference against itself (8 vs 3 threads), last step's update, relative L2:", floor, "branches", br_a, br_b)
    save("g15_warm_step", names=np.array(names), update_8=np.stack(upd_a), update_3=np.stack(upd_b), pre_8=np.stack(pre),
         noise_floor=np.array(floor), logs_8=np.array(json.dumps(logs_a)), logs_3=np.array(json.dumps(logs_b)),
         branch=np.array(br_a), cfg=np.array(json.dumps(G15)))


def g14_config(cfg, spec):
    """The settings G14 adds to make_cfg's (shared with the tests that replay it: tests/test_oracle_golden.py,
    tests/test_hip_model.py): a 4-sample monitor window, so that the trend `dev_avg` -- zero until the window is full
    (monitoring.py:64-73) -- is live from the 4th step and feeds both model_select and evaluate_update_dynamic within
    two 3-step domains; a replay-buffer fill rate that makes buffer_update fire; samples at every epoch end; and a
    learning rate 50x below the yml's: a train-mode step on random weights amplifies rounding differences ~40x per step
    (the reference's own second update moves by 19 % with its thread count), so at the yml's rate a
    six-step trajectory would pin nothing after its third step -- at this rate the whole stream is comparable."""
    spec.AVG_MONITOR_SIZE = G14["monitor"]
    spec.LEARNING_RATE = G14["lr"]
    spec.EPOCHS = 1
    cfg.TRAINING.PERC_FILL_PER_DOMAIN = G14["perc_fill"]
    cfg.OTHERS.GENERATE_SAMPLES_EVERY = 3
    cfg.classnum_to_label = {i: str(i) for i in range(19)}
    cfg.device = cfg.OTHERS.DEVICE


def g14():
    """The OUTER loop the driver calls (train_ouda.py:227-261): two target "domains", each `update_cfg_spec` +
    `hybrid_proDA.train(src_loader, trg_loader, val_set)` (prototypes.py:466-520) over list-backed synthetic loaders at
    128x64, batch 2 -- 1 epoch x 3 steps per domain, one validation set of 10 frames.  Domain 0 as a first domain runs
    (SKIP_CALC False: prototypes from the source loader + an initial evaluation; AUTO_DYNAMIC unset: the dynamic model
    is refreshed at the start); domain 1 as every later one (SKIP_CALC True, and AUTO_DYNAMIC True through
    ORDER_OPTIONS: no refresh at the start, `evaluate_update_dynamic` decides -- its 500-step counter is pre-set to 499
    between the domains so that the decision falls inside the run).  Captured: every dict handed to wandb.log (scalars;
    sample images as their class maps), the replay-buffer additions, prototypes, the switch trajectory, digests of all
    four models at the end of each domain."""
    import warnings
    import wandb
    from onda_amd.synthetic import ListLoader
    logs = []

    class Image:  # the stub's wandb.Image, keeping what the reference hands it (framework/utils/logging.py:13-17)
        def __init__(self, data, masks=None, caption=None):
            self.pred = np.asarray(masks["predictions"]["mask_data"]).astype(np.uint8)

    wandb.Image = Image
    wandb.log = lambda d, *a, **k: logs.append(dict(d))
    res = {}
    with tempfile.TemporaryDirectory() as tmp:
        cfg, spec = make_cfg(tmp)
        g14_config(cfg, spec)
        model = ref_model(1, G14["head_scale"])
        da = hybrid_proDA(model, cfg, spec)
        n = G14["steps"]
        src = ListLoader([synth_batch(2, 64, 128, seed=1400 + i) for i in range(2)])
        domains = [ListLoader([synth_batch(2, 64, 128, seed=1500 + 10 * d + i) for i in range(n)]) for d in range(2)]
        val = {"val": ListLoader([synth_batch(1, 64, 128, seed=1600 + i) for i in range(G14["val_frames"])])}
        torch.manual_seed(123)
        np.random.seed(G14["np_seed"])
        branch = []
        step0 = da.step

        def step(*a, **k):  # (records the switch after every step; the step itself is the reference's)
            out = step0(*a, **k)
            branch.append([da.model_select.current, da.model_select.current_dev])
            return out
        da.step = step
        f_domain = False
        for d, (set_, loader) in enumerate(zip(((25,), (50,)), domains)):
            spec.set_ = set_
            if d == 1:
                spec["AUTO_DYNAMIC"] = True           # train_ouda.py:252-256 (SCHEME.ORDER_OPTIONS)
                da.dynamic_update_counter = 499       # (see the docstring)
            spec.SKIP_CALC |= f_domain                # train_ouda.py:257-258
            f_domain = True
            da.update_cfg_spec(spec)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                da.train(src, loader, val)
            res[f"proto_d{d}"], res[f"sqmean_d{d}"] = da.prototypes.prototypes.clone(), da.prototypes.squared_mean.clone()
            res[f"counter_d{d}"] = da.prototypes.counter.clone()
            res[f"dynamic_counter_d{d}"] = np.array(da.dynamic_update_counter)
            names, dig = [], []
            for who, mod in (("student.", da.model), ("teacher.", da.ema_model), ("dynamic.", da.dynamic_model),
                             ("static.", da.static_model)):
                for k_, p in mod.state_dict().items():
                    names.append(who + k_)
                    dig.append(digest(p.float(), 64))
            res[f"state_names_d{d}"], res[f"state_digest_d{d}"] = np.array(names), np.stack(dig)
            res[f"files_d{d}"] = np.array(sorted(os.listdir(tmp)))
            if d == 0:
                from framework.domain_adaptation.methods.prototype_handler import prototype_handler as ph
                chk = ph(0.9995, 1, 0.3, "mahalanobis")
                assert chk.load(os.path.join(tmp, "proto_(25,).pickle"))
        res["branch"] = np.array(branch)
        scalars, nimg = [], 0
        for i, d in enumerate(logs):
            row = {}
            for k_, v in d.items():
                if isinstance(v, Image):
                    res[f"log{i}_img_{k_}"] = v.pred
                    nimg += 1
                else:
                    v = tolog({k_: v})[k_]
                    if np.isscalar(v):
                        row[k_] = v
            scalars.append(row)
        res["logs_json"] = np.array(json.dumps(scalars))
        res["added_index"] = np.array([i for i, _ in src.added], dtype=np.int64)
        res["added_maps"] = np.stack([m.numpy() for _, m in src.added]) if src.added else np.zeros((0, 64, 128), np.uint8)
        res["added_at"] = np.array([i for i, row in enumerate(scalars) if row.get("Total buffer updates", 0) > 0])
        save("g14_train_loop", **res)
        print("g14:", len(logs), "log dicts,", nimg, "sample maps,", len(src.added), "buffer additions, branch", branch,
              "dynamic counters", res["dynamic_counter_d0"], res["dynamic_counter_d1"])
        for i, row in enumerate(scalars):
            print(i, {k_: round(v, 5) for k_, v in row.items() if k_ in ("Total target loss", "prior static confidence ma", "dev avg prior static", "Total buffer updates", "Val mIoU model of val", "pseudolabel_pixel_num", "buff_loss")})


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g5", "g6", "g7", "g8", "g9", "g10", "g11", "g12", "g13", "g14", "g15"]
    for w in which:
        globals()[w]()
