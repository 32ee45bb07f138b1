"""GPU parity of the drop-in model and of one full adaptation step, against the golden vectors
captured from the reference (G1, G2, G7) and against the CPU oracle on the same inputs."""
import json
import os
from copy import deepcopy

import numpy as np
import pytest
import torch

from conftest import digest

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(autouse=True, params=["f32", "f16x2"])
def conv_mode(request):
    """Every model-level parity test runs with both conv evaluations: exact fp32 MFMA and the two-limb f16 split
    (fp32-level accuracy on the f16 pipe)."""
    from onda_amd import ops
    old, ops.CONV_MODE = ops.CONV_MODE, request.param
    yield request.param
    ops.CONV_MODE = old


def build_model(seed, head_scale):
    from onda_amd.framework.model.deeplabv2 import get_deeplab_v2
    from onda_amd.synthetic import fill_state_dict
    m = get_deeplab_v2(num_classes=19, multi_level=True, layers=[3, 4, 6, 3], classifier="ProDA")
    m.multi_level = False
    fill_state_dict(m, seed, head_scale)
    return m.to(DEV)


def _log_close(mine, ref, key, step, npix):
    """Log scalars: 5e-3 relative.  Pixel-count ratios of the SECOND step (agreement / kept fractions
    over the 306 target pixels) sit behind one SGD step through train-mode BN, where the reference's own
    weights move by 19 % under a thread-count change (see the conditioning notes above): a single pixel
    changing side of an argmax tie is within the reference's own noise, so one pixel is allowed there."""
    if mine == pytest.approx(ref, rel=5e-3, abs=1e-5):
        return True
    if step >= 1 and ("agreement" in key or "percentage" in key or "pixel_num" in key):
        one = 1.0 if "pixel_num" in key else 1.0 / npix
        return abs(mine - ref) <= 1.01 * one
    return False



def test_eval_forward_small_golden(golden):
    """Eval-mode forward (folded BN): logits within 1e-3 rel of the reference, argmax map of
    interp(out) bit-identical (BASELINE north_star)."""
    from onda_amd import ops
    from onda_amd.synthetic import synth_batch
    g = golden("g1_eval_small")
    m = build_model(1, 3.0).eval()
    x = synth_batch(2, 64, 128, seed=7)["image"].to(DEV)
    with torch.no_grad():
        x1, o = m(x)
        cls = ops.upsample_argmax(o["out"], (64, 128))
        up = ops.UpsampleFn.apply(o["out"], (64, 128))
    assert x1 is None and o["feat"].shape == (2, 256, 9, 17) and o["out"].shape == (2, 19, 9, 17)
    for key, tol in (("feat", 1e-4), ("out", 1e-4), ):
        ref = g[key]
        err = np.abs(o[key].cpu().numpy() - ref).max()
        assert err <= tol * np.abs(ref).max(), (key, err)
    assert np.abs(up.cpu().numpy() - g["up"]).max() <= 1e-4 * np.abs(g["up"]).max()
    assert np.array_equal(cls.cpu().numpy(), g["argmax"])


def test_eval_forward_512x1024_golden(golden):
    """One full-resolution frame (config 1 shape): class map vs the reference's."""
    from onda_amd import ops
    from onda_amd.synthetic import synth_batch
    g = golden("g1_eval_large")
    m = build_model(1, 3.0).eval()
    x = synth_batch(1, 512, 1024, seed=8)["image"].to(DEV)
    with torch.no_grad():
        _, o = m(x)
        cls = ops.upsample_argmax(o["out"], (512, 1024)).cpu().numpy()
    assert o["out"].shape == (1, 19, 65, 129)
    ref = g["out"]
    assert np.abs(o["out"].cpu().numpy() - ref).max() <= 1e-3 * np.abs(ref).max()
    diff = cls != g["argmax"]
    # any disagreement must sit on a numerical tie of the reference's own top-2 logits
    assert diff.mean() < 1e-5
    if diff.any():
        assert g["margin_f16"].astype(np.float32)[diff].max() < 1e-3
    np.testing.assert_allclose(digest(o["feat"], 4096)[2:], g["feat_digest"][2:], rtol=0,
                               atol=1e-3 * np.abs(g["feat_digest"][2:]).max())


@pytest.mark.parametrize("track", [True, False])
def test_train_forward_backward_golden(golden, track):
    """Train-mode pass (batch-statistics BN, fixed Dropout2d mask): logits within 1e-3 rel,
    gradients of all 82 trainable tensors and running statistics against the reference (G2)."""
    from onda_amd import ops
    from onda_amd.framework.model import deeplabv2
    from onda_amd.framework.domain_adaptation.methods.adaptation_model import switch_batch_statistics
    from onda_amd.synthetic import synth_batch
    g = golden("g2_train_small")
    tag = "track" if track else "frozen"
    m = build_model(1, 3.0).train()
    switch_batch_statistics(m, track)
    b = synth_batch(2, 64, 128, seed=7)
    mask = torch.from_numpy(g["drop_mask"])
    deeplabv2.drop_mask_fn = lambda B, C, p, dev: mask.to(dev)
    try:
        _, o = m(b["image"].to(DEV))
    finally:
        deeplabv2.drop_mask_fn = deeplabv2._default_drop_mask
    loss = ops.seg_losses(o["out"], b["label_res"].to(DEV), 1.0, 0.0, 0.0)[0]
    ref = g[f"{tag}_out"]
    assert np.abs(o["out"].detach().cpu().numpy() - ref).max() <= 1e-3 * np.abs(ref).max()
    assert np.abs(o["feat"].detach().cpu().numpy() - g[f"{tag}_feat"]).max() <= 1e-3 * np.abs(g[f"{tag}_feat"]).max()
    assert abs(loss.item() - g[f"{tag}_loss"]) <= 1e-4 * abs(g[f"{tag}_loss"])
    sd = m.state_dict()
    for k in ("bn1", "layer1.0.bn1", "layer2.0.downsample.1", "layer3.2.bn2", "layer4.2.bn3"):
        np.testing.assert_allclose(sd[k + ".running_mean"].cpu().numpy(), g[f"{tag}_{k}.running_mean"], rtol=1e-3, atol=1e-5)
        np.testing.assert_allclose(sd[k + ".running_var"].cpu().numpy(), g[f"{tag}_{k}.running_var"], rtol=1e-3, atol=1e-5)
        assert int(sd[k + ".num_batches_tracked"]) == int(g[f"{tag}_{k}.num_batches_tracked"])
    if not track:
        return
    loss.backward()
    params = dict(m.named_parameters())
    names = list(g["grad_names"])
    assert sorted(names) == sorted(n for n, p in params.items() if p.grad is not None)
    # Train-mode BN on random weights amplifies fp32 rounding: the reference's own fp32 gradients
    # sit up to 4 % (sampled max-norm) from an fp64 evaluation and move by 1 % when only the CPU
    # thread count changes (DESIGN.md "conditioning").  So each gradient is held to the fp64
    # oracle with a budget of 3x the reference's own fp32 distance from it.
    ref64 = _fp64_oracle_grads(g, b, mask, names)
    for n, dg in zip(names, g["grad_digest"]):
        mine, r64 = digest(params[n].grad)[2:], ref64[n][2:]
        norm = np.linalg.norm(r64) + 1e-30
        e_mine, e_ref = np.linalg.norm(mine - r64) / norm, np.linalg.norm(dg[2:] - r64) / norm
        # the floor (1e-2) is the size of the reference's own run-to-run movement: its gradients move
        # by up to 1 % when only the CPU thread count changes, and this path's two conv evaluations
        # (exact fp32 MFMA vs 3-limb bf16 split, 2e-7 apart per conv) differ by 0.4 % median here
        assert e_mine <= 3 * e_ref + 1e-2, (n, e_mine, e_ref)
    for n in ("layer6.head.1.weight", "layer6.bottleneck.2.weight", "layer6.conv2d_list.0.0.bias"):
        ref = g["grad_" + n]  # close to the loss: well conditioned
        assert np.abs(params[n].grad.cpu().numpy() - ref).max() <= 5e-3 * np.abs(ref).max(), n


def _fp64_oracle_grads(g, batch, mask, names):
    from oracle import losses, model as omodel
    from onda_amd.synthetic import synth_tensor
    sd = {k: synth_tensor(k, torch.empty(shape, dtype=dt), 1, 3.0).to(dt) for k, shape, dt in omodel.state_spec()}
    sd = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    for k in names:
        sd[k].requires_grad_(True)
    _, o = omodel.forward(batch["image"].double(), sd, omodel.BNMode(True, True, 0.1), mask.double())
    loss = losses.ce_hard(o["out"], batch["label_res"])
    grads = torch.autograd.grad(loss, [sd[n] for n in names])
    return {n: digest(gr) for n, gr in zip(names, grads)}


def test_model_invariants_on_device():
    """What the reference's callers rely on: deepcopy, per-module BN state_dict,
    376-key state_dict round trip, BatchNorm2d instances, param/buffer order."""
    import torch.nn as nn
    m = build_model(2, 1.0)
    c = deepcopy(m)
    sd = m.state_dict()
    assert len(sd) == 376 and list(sd.keys()) == list(c.state_dict().keys())
    assert sum(isinstance(x, nn.BatchNorm2d) for x in m.modules()) == 53
    assert len(list(m.parameters())) == 217 and len(list(m.buffers())) == 159
    x = torch.randn(1, 3, 64, 128, device=DEV)
    m.eval(); c.eval()
    with torch.no_grad():
        a, b = m(x)[1]["out"], c(x)[1]["out"]
        assert torch.equal(a, b)
        c.load_state_dict({k: v * 1.01 if v.dtype == torch.float32 else v for k, v in sd.items()})
        assert not torch.equal(c(x)[1]["out"], a)  # packed weights follow the parameters
        c.load_state_dict(sd)
        assert torch.equal(c(x)[1]["out"], a)
    with pytest.raises(RuntimeError):
        m.cpu()(x.cpu())  # no CPU fallback


@pytest.mark.parametrize("multi_level", [False, True])
def test_shared_gradient_buffers_match_autograd_accumulation(multi_level):
    """ops.GradSink (gradients of a block input / of the ASPP input accumulated in place by the data-gradient kernels)
    against autograd's own sum of the consumers' gradients: same parameter gradients, same input-side gradients, on a
    ragged size, and a second backward pass through a fresh graph starts from a clean buffer."""
    from onda_amd import ops
    m = build_model(4, 3.0).train()
    m.multi_level = multi_level
    torch.manual_seed(3)
    x = torch.randn(2, 3, 97, 161, device=DEV)
    lab = torch.randint(0, 19, (2, 13, 21), device=DEV)

    def grads(share):
        old, ops.SHARE_GRADS = ops.SHARE_GRADS, share
        try:
            m.zero_grad(set_to_none=True)
            from onda_amd.framework.model import deeplabv2
            deeplabv2.force_mask(torch.ones(2, 256, device=DEV))
            if multi_level:
                deeplabv2.force_mask(torch.ones(2, 256, device=DEV))
            o1, o2 = m(x)
            loss = ops.seg_losses(o2["out"], lab, 1.0, 0.0, 0.0)[0]
            if multi_level:
                loss = loss + 0.1 * ops.seg_losses(o1["out"], lab, 1.0, 0.0, 0.0)[0]
            loss.backward()
            return {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
        finally:
            ops.SHARE_GRADS = old

    ref = grads(False)
    for _ in range(2):
        got = grads(True)
        assert got.keys() == ref.keys() and len(ref) > 50
        for n in ref:
            scale = ref[n].abs().max().item() + 1e-30
            assert (got[n] - ref[n]).abs().max().item() <= 2e-5 * scale, n


@pytest.mark.parametrize("tag,head_scale", [("static", 40.0), ("dynamic", 3.0)])
def test_full_step_golden(golden, tmp_path, tag, head_scale):
    """Two complete hybrid_proDA steps (+update_ema) at 128x64, B=2 against the reference's log
    dict, labels, prototypes and post-step weights (fixture G7), both switch branches."""
    from onda_amd.config import hybrid_switch_cfg
    from onda_amd.framework.handlers import get_adapt_method, get_model
    from onda_amd.framework.model import deeplabv2
    from onda_amd.framework.domain_adaptation.methods.adaptation_model import switch_batch_statistics
    from onda_amd.synthetic import fill_state_dict, synth_batch
    from oracle import model as omodel
    g = golden(f"g7_step_{tag}")
    cfg, spec = hybrid_switch_cfg(128, 64, DEV, str(tmp_path), batch_size=2)
    model = get_model(cfg, 19)
    fill_state_dict(model, 1, head_scale)
    da = get_adapt_method(cfg)(model, cfg, spec)
    src = [synth_batch(2, 64, 128, seed=100 + i) for i in range(2)]
    trg = [synth_batch(2, 64, 128, seed=200 + i) for i in range(2)]
    torch.manual_seed(123)
    masks = [omodel.draw_drop_mask(2) for _ in range(8)]  # same CPU draws as the reference run
    it = iter(masks)
    deeplabv2.drop_mask_fn = lambda B, C, p, dev: next(it).to(dev)
    try:
        da.update_dynamic()
        switch_batch_statistics(da.model, False)
        da.calculate_prototypes(src, save=False)
        switch_batch_statistics(da.model, True)
        np.testing.assert_allclose(da.prototypes.prototypes.cpu().numpy(), g["proto0"], rtol=1e-3, atol=1e-4)
        np.testing.assert_allclose(da.prototypes.counter.cpu().numpy(), g["counter0"])
        da.optimizer.zero_grad()
        prev_digest = {who + k: digest(v.float(), 64)[2:] for who, mod in (("student.", da.model), ("teacher.", da.ema_model))
                       for k, v in mod.state_dict().items()}
        for s in range(2):
            da.adjust_learning_rate(s, 6)
            log = da.step([src[s]], trg[s])
            da.update_ema()
            # the log leaves the step as plain values: keeping it must not keep the step's autograd graph alive
            assert not any(torch.is_tensor(v) and (v.requires_grad or v.grad_fn is not None) for v in log.values())
            assert int(g[f"branch{s}"]) == da.model_select.current
            soft = trg[s]["stored_predictions"]
            assert (soft.cpu() - torch.from_numpy(g[f"soft{s}"])).abs().max() < 2e-3
            ref = json.loads(str(g[f"log{s}_json"]))
            for k, v in ref.items():
                mine = log[k]
                mine = mine.item() if isinstance(mine, torch.Tensor) else float(mine)
                assert _log_close(mine, v, k, s, soft[0, 0].numel() * soft.shape[0]), (s, k, mine, v)
            np.testing.assert_allclose(da.prototypes.prototypes.cpu().numpy(), g[f"proto{s + 1}"], rtol=1e-3, atol=1e-4)
            # Post-step weights, compared as UPDATES (w_after - w_before).  The reference's own
            # update moves by 0.3 % (step 0) and 19 % (step 1, chaotic amplification through two
            # train-mode passes) in relative L2 when only its CPU thread count changes
            # (DESIGN.md "conditioning"), which bounds what parity can mean here.
            names, dg = list(g[f"state_names{s}"]), g[f"state_digest{s}"]
            num = den = 0.0
            for who, mod in (("student.", da.model), ("teacher.", da.ema_model)):
                for k, v in mod.state_dict().items():
                    if not v.is_floating_point() or v.dim() == 0:
                        continue
                    row = dg[names.index(who + k)][2:]
                    before = prev_digest[who + k]
                    mine = digest(v.float(), 64)[2:]
                    num += ((mine - row) ** 2).sum()
                    den += ((row - before) ** 2).sum()
                    prev_digest[who + k] = row
            assert (num / den) ** 0.5 <= (0.02 if s == 0 else 0.6), (s, (num / den) ** 0.5)
    finally:
        deeplabv2.drop_mask_fn = deeplabv2._default_drop_mask


def test_step_at_the_yml_learning_rate_from_a_warm_state(golden, tmp_path, conv_mode):
    """Fixture G15: three steps at a learning rate 50x below the yml's (where G14 pins whole trajectories), then ONE step at
    the yml's own rate -- the step the trajectory fixtures leave open.  The reference ran the sequence twice, with 8 and
    with 3 CPU threads; the distance between ITS OWN two updates (9.6 % in relative L2: a train-mode step on random weights
    amplifies summation-order noise) is the noise floor this step has in the reference itself.  The HIP step has to land
    within 2x that floor of BOTH captures -- a bound taken from the reference, not from comparing this repo with itself."""
    from onda_amd.config import hybrid_switch_cfg
    from onda_amd.framework.handlers import get_adapt_method, get_model
    from onda_amd.framework.model import deeplabv2
    from onda_amd.framework.domain_adaptation.methods.adaptation_model import switch_batch_statistics
    from onda_amd.synthetic import fill_state_dict, synth_batch
    from oracle import model as omodel
    g = golden("g15_warm_step")
    setup = json.loads(str(g["cfg"]))
    cfg, spec = hybrid_switch_cfg(128, 64, DEV, str(tmp_path), batch_size=2)
    model = get_model(cfg, 19)
    fill_state_dict(model, 1, 40.0)
    da = get_adapt_method(cfg)(model, cfg, spec)
    lr, n = float(spec.LEARNING_RATE), setup["warm_steps"] + 1
    src = [synth_batch(2, 64, 128, seed=500 + i) for i in range(n)]
    trg = [synth_batch(2, 64, 128, seed=600 + i) for i in range(n)]
    torch.manual_seed(123)
    masks = [omodel.draw_drop_mask(2) for _ in range(2 + 3 * n)]  # the reference's CPU draws, in its order (as test_full_step_golden)
    it = iter(masks)
    deeplabv2.drop_mask_fn = lambda B, C, p, dev: next(it).to(dev)
    try:
        da.update_dynamic()
        switch_batch_statistics(da.model, False)
        da.calculate_prototypes(src[:2], save=False)
        switch_batch_statistics(da.model, True)
        da.optimizer.zero_grad()
        logs = []
        for s_ in range(n):
            da.cfg_spec.LEARNING_RATE = lr / setup["warm_lr_div"] if s_ < setup["warm_steps"] else lr
            da.adjust_learning_rate(s_, 8)
            if s_ == n - 1:
                before = {k: v.detach().double().cpu().clone() for k, v in da.model.state_dict().items()}
            log = da.step([src[s_]], trg[s_])
            da.update_ema()
            # (read now: the monitor's moving averages in a step's log are evaluated when they are looked at)
            logs.append({k: (v.item() if isinstance(v, torch.Tensor) else float(v)) for k, v in log.items()
                         if not isinstance(v, dict) and (not isinstance(v, torch.Tensor) or v.numel() == 1)})
    finally:
        deeplabv2.drop_mask_fn = deeplabv2._default_drop_mask
    assert int(g["branch"]) == da.model_select.current
    ref_logs = json.loads(str(g["logs_8"]))
    for s_ in range(n):  # the warm steps and the last step's log: the usual 5e-3 (growing with the step as in G14)
        for k, v in ref_logs[s_].items():
            mine = logs[s_][k]
            mine = mine.item() if isinstance(mine, torch.Tensor) else float(mine)
            assert mine == pytest.approx(v, rel=5e-3 * 1.5 ** s_, abs=2e-5) or ("pixel_num" in k and abs(mine - v) <= 2) or \
                (("agreement" in k or "percentage" in k) and abs(mine - v) <= 2.01 / 306), (s_, k, mine, v)  # (a pixel or two of 306 at a tie)
    names = list(g["names"])
    after = da.model.state_dict()
    floor = float(g["noise_floor"])
    dev = {}
    for tag in ("update_8", "update_3"):
        num = den = 0.0
        for i, k in enumerate(names):
            mine = digest(after[k].detach().double().cpu() - before[k], 64)[2:]
            row = g[tag][i][2:]
            num += ((mine - row) ** 2).sum()
            den += (row ** 2).sum()
        dev[tag] = (num / den) ** 0.5
    print(f"g15: update vs the reference's 8-thread / 3-thread runs {dev['update_8']:.4f} / {dev['update_3']:.4f}; "
          f"the reference against itself {floor:.4f}")
    assert 0.02 < floor < 0.3  # (the fixture's own sanity: this step IS noisy in the reference, and not garbage)
    assert max(dev.values()) <= 2.0 * floor, (dev, floor)


def _full_size_step_against(golden, tmp_path, name, width, height, batch, head_scale, seeds=(1000, 2000)):
    """ONE hybrid_proDA step (+update_ema) at a full size against the reference's run of the same step (fixtures G10 / G12 /
    G13, tests/golden/make_golden.py::_full_step): branch, log dict, pseudo-label map (outside the reference's own
    numerical ties), prototypes before / after, post-step weight updates."""
    from onda_amd import ops
    from onda_amd.config import hybrid_switch_cfg
    from onda_amd.framework.handlers import get_adapt_method, get_model
    from onda_amd.framework.model import deeplabv2
    from onda_amd.framework.domain_adaptation.methods.adaptation_model import switch_batch_statistics
    from onda_amd.synthetic import fill_state_dict, synth_batch
    from oracle import model as omodel
    g = golden(name)
    cfg, spec = hybrid_switch_cfg(width, height, DEV, str(tmp_path), batch_size=batch)
    model = get_model(cfg, 19)
    fill_state_dict(model, 1, head_scale)
    da = get_adapt_method(cfg)(model, cfg, spec)
    src = [synth_batch(batch, height, width, seed=seeds[0] + i) for i in range(2)]
    trg = synth_batch(batch, height, width, seed=seeds[1])
    torch.manual_seed(123)
    masks = iter([omodel.draw_drop_mask(batch) for _ in range(6)])  # same CPU draws as the reference run
    deeplabv2.drop_mask_fn = lambda B, C, p, dev: next(masks).to(dev)
    try:
        da.update_dynamic()
        switch_batch_statistics(da.model, False)
        da.calculate_prototypes(src, save=False)
        switch_batch_statistics(da.model, True)
        np.testing.assert_allclose(da.prototypes.prototypes.cpu().numpy(), g["proto0"], rtol=1e-3, atol=1e-4)
        np.testing.assert_allclose(da.prototypes.counter.cpu().numpy(), g["counter0"])
        da.optimizer.zero_grad()
        before = {who + k: digest(v.float(), 64)[2:] for who, mod in (("student.", da.model), ("teacher.", da.ema_model))
                  for k, v in mod.state_dict().items()}
        da.adjust_learning_rate(0, 6)
        log = da.step([src[0]], trg)
        da.update_ema()
    finally:
        deeplabv2.drop_mask_fn = deeplabv2._default_drop_mask
    assert int(g["branch"]) == da.model_select.current
    soft = trg["stored_predictions"].float().cpu()
    labels = soft.argmax(1).to(torch.uint8).numpy()
    tie = np.unpackbits(g["tie_mask"])[: labels.size].reshape(labels.shape).astype(bool)
    wrong = (labels != g["labels"]) & ~tie
    assert wrong.sum() == 0, int(wrong.sum())
    # the tie mask (pixels whose two best soft probabilities are within 2e-3 in the reference's own run) must not absorb a
    # regression: only a small part of the masked pixels may actually come out differently
    inside = int(((labels != g["labels"]) & tie).sum())
    print(f"{name}: {int(tie.sum())} of {labels.size} pixels in the reference's tie mask ({tie.mean():.4%}), {inside} of them differ")
    assert inside <= max(2, 0.02 * tie.sum()), (inside, int(tie.sum()))  # (measured: none of G10 / G12 / G13's masked pixels differ)
    assert np.abs(soft.max(1)[0].numpy() - g["soft_max"].astype(np.float32)).max() < 3e-3
    for k, v in json.loads(str(g["log_json"])).items():
        mine = log[k]
        mine = mine.item() if isinstance(mine, torch.Tensor) else float(mine)
        assert _log_close(mine, v, k, 0, labels.size), (k, mine, v)
    np.testing.assert_allclose(da.prototypes.prototypes.cpu().numpy(), g["proto1"], rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(da.prototypes.squared_mean.cpu().numpy(), g["sqmean1"], rtol=1e-3, atol=1e-4)
    names, dg = list(g["state_names"]), g["state_digest"]
    num = den = 0.0
    for who, mod in (("student.", da.model), ("teacher.", da.ema_model)):
        for k, v in mod.state_dict().items():
            if not v.is_floating_point() or v.dim() == 0:
                continue
            row = dg[names.index(who + k)][2:]
            num += ((digest(v.float(), 64)[2:] - row) ** 2).sum()
            den += ((row - before[who + k]) ** 2).sum()
    print(f"{name} [{ops.CONV_MODE}]: post-step weights as updates, relative L2 against the reference {(num / den) ** 0.5:.5f}")
    assert (num / den) ** 0.5 <= 0.02, (num / den) ** 0.5  # post-step weights, as updates (see test_full_step_golden)
    return tie.mean()


def test_full_size_step_golden(golden, tmp_path, conv_mode):
    """BASELINE config 3 on the STATIC side of the switch (head x40): 512x1024, batch 4 (fixture G10) -- in BOTH conv modes
    since round 5 (the exact-fp32 kernels take 4 s for it; measured deviations: profiles/r05_g10_exact_f32_mode.txt)."""
    _full_size_step_against(golden, tmp_path, "g10_step_full", 1024, 512, 4, 40.0)


def test_bench_step_golden(golden, tmp_path, conv_mode):
    """The step bench.py TIMES, in the state it times it in: 512x1024, batch 4, head x1 = the DYNAMIC branch, source seeds
    1000 / 1001, target seed 2000 (fixture G12, captured from the imported reference)."""
    if conv_mode != "f16x2":
        pytest.skip("full-size step: default conv mode only (the small-size step runs in both)")
    _full_size_step_against(golden, tmp_path, "g12_step_bench_dynamic", 1024, 512, 4, 1.0)
    assert int(golden("g12_step_bench_dynamic")["branch"]) == 1


@pytest.mark.parametrize("batch,name", [(2, "g13_step_1024x2048"), (4, "g13_step_1024x2048_bs4")])
def test_full_resolution_step_golden(golden, tmp_path, conv_mode, batch, name):
    """One adaptation step at BASELINE config 5's resolution, 1024x2048 (feature grid 129x257): at batch 2 (fixture G13) and at
    batch 4 -- config 5's own batch, the one bench.py's config-5 line runs; the reference's CPU run of the batch-4 step takes
    30 GB and eight minutes in the build container (make_golden.py::g13b).  Since round 5 the student's two passes pair at
    this size as well (4 + 4 images: layer4's 2048-channel limb rows are 2.17 GB -- the kernels address an operand through
    per-tile windows, csrc/conv_l2.hip x_window)."""
    if conv_mode != "f16x2":
        pytest.skip("full-size step: default conv mode only (the small-size step runs in both)")
    _full_size_step_against(golden, tmp_path, name, 2048, 1024, batch, 40.0, seeds=(1300, 2300))


def test_f16x2_steps_track_the_exact_f32_steps(tmp_path, conv_mode):
    """The default arithmetic against this repo's OWN exact-fp32 kernels through two complete adaptation steps (the G7
    recipe, dynamic side): same branch, same labels, logs 1e-4, and the post-step weights -- as UPDATES -- within 1e-3 after
    BOTH steps.  (Against the reference the second step's update is only held to 60 %: its own update moves by 19 % with its
    CPU thread count.  Here both sides are deterministic and differ only in the conv arithmetic, so the bound is tight.)"""
    if conv_mode != "f16x2":
        pytest.skip("compares the two modes itself")
    from onda_amd import ops
    from onda_amd.config import hybrid_switch_cfg
    from onda_amd.framework.handlers import get_adapt_method, get_model
    from onda_amd.framework.model import deeplabv2
    from onda_amd.framework.domain_adaptation.methods.adaptation_model import switch_batch_statistics
    from onda_amd.synthetic import fill_state_dict, synth_batch
    from oracle import model as omodel

    def run(mode):
        old, ops.CONV_MODE = ops.CONV_MODE, mode
        try:
            cfg, spec = hybrid_switch_cfg(128, 64, DEV, str(tmp_path / mode), batch_size=2)
            model = get_model(cfg, 19)
            fill_state_dict(model, 1, 3.0)
            da = get_adapt_method(cfg)(model, cfg, spec)
            src = [synth_batch(2, 64, 128, seed=100 + i) for i in range(2)]
            trg = [synth_batch(2, 64, 128, seed=200 + i) for i in range(2)]
            torch.manual_seed(123)
            it = iter([omodel.draw_drop_mask(2) for _ in range(8)])
            deeplabv2.drop_mask_fn = lambda B, C, p, dev: next(it).to(dev)
            da.update_dynamic()
            switch_batch_statistics(da.model, False)
            da.calculate_prototypes(src, save=False)
            switch_batch_statistics(da.model, True)
            da.optimizer.zero_grad()
            start = {k: v.detach().clone() for k, v in da.model.state_dict().items() if v.is_floating_point() and v.dim() > 0}
            out = []
            for s in range(2):
                da.adjust_learning_rate(s, 6)
                log = da.step([src[s]], trg[s])
                da.update_ema()
                out.append(({k: (v.item() if torch.is_tensor(v) else float(v)) for k, v in log.items()},
                            trg[s]["stored_predictions"].argmax(1).cpu(), da.model_select.current,
                            {k: v.detach().clone() for k, v in da.model.state_dict().items() if k in start}))
            return start, out
        finally:
            ops.CONV_MODE = old
            deeplabv2.drop_mask_fn = deeplabv2._default_drop_mask

    start, a = run("f16x2")
    _, b = run("f32")
    prev, seen = start, []
    for s in range(2):
        (la, laba, bra, wa), (lb, labb, brb, wb) = a[s], b[s]
        assert bra == brb
        worst = max((abs(la[k] - v) / max(abs(v), 1e-6), k) for k, v in lb.items() if np.isfinite(v))
        num = sum(((wa[k] - wb[k]).double() ** 2).sum().item() for k in wb)
        den = sum(((wb[k] - prev[k]).double() ** 2).sum().item() for k in wb)
        seen.append({"step": s, "worst_log_rel": worst[0], "worst_log_key": worst[1], "update_rel_l2": (num / den) ** 0.5,
                     "labels_differ": int((laba != labb).sum())})
        prev = wb
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "f16x2_vs_f32_steps.json"), "w") as f:
            json.dump(seen, f)
    # step 0 starts from identical weights: the two arithmetics differ by their own rounding only (measured: logs 1e-6,
    # update 1.2e-4).  Step 1 runs on weights that already differ by that much, seen through a train-mode pass on random
    # weights -- the conditioning that makes the REFERENCE's own second update move by 19 % with its thread count --:
    # measured logs 1.0e-4, update 4.6e-3 (a factor 38 in one step; the reference's 60 % tolerance there is that factor
    # applied to ITS fp32 noise)
    assert seen[0]["worst_log_rel"] <= 1e-4 and seen[0]["update_rel_l2"] <= 1e-3 and seen[0]["labels_differ"] == 0, seen
    assert seen[1]["worst_log_rel"] <= 1e-3 and seen[1]["update_rel_l2"] <= 2e-2, seen


def test_a_step_never_stops_the_host(tmp_path, conv_mode):
    """With the switch on the device NOTHING in an adaptation step waits for the GPU: behind ~1.5 s of queued device work
    the host gets through step() + update_ema() in a fraction of the time that work takes to drain (the host-decided
    switch -- ONDA_DEVICE_SWITCH=0, every other conv mode -- has to wait for the static model's confidence there).  Also:
    both sides of the switch through the device path, static (predicated dynamic pass returns at once) and dynamic."""
    if conv_mode != "f16x2":
        pytest.skip("the device-side switch runs with the default (pre-split) kernels")
    import time
    from onda_amd.config import hybrid_switch_cfg
    from onda_amd.framework.handlers import get_adapt_method, get_model
    from onda_amd.framework.domain_adaptation.methods.adaptation_model import switch_batch_statistics
    from onda_amd.synthetic import fill_state_dict, synth_batch
    for head, branch in ((40.0, 0), (1.0, 1)):
        cfg, spec = hybrid_switch_cfg(256, 128, DEV, str(tmp_path), batch_size=2)
        model = get_model(cfg, 19)
        fill_state_dict(model, 1, head)
        da = get_adapt_method(cfg)(model, cfg, spec)
        assert da._dsw is not None and da.model_select.device_switch is da._dsw
        src = [{k: v.to(DEV) for k, v in synth_batch(2, 128, 256, seed=100 + i).items()} for i in range(2)]
        trg = [{k: v.to(DEV) for k, v in synth_batch(2, 128, 256, seed=200 + i).items()} for i in range(2)]
        da.update_dynamic()
        switch_batch_statistics(da.model, False)
        da.calculate_prototypes(src, save=False)
        switch_batch_statistics(da.model, True)
        da.optimizer.zero_grad()
        logs = []
        for i in range(3):  # (allocator, pinned buffers and kernels warm)
            da.adjust_learning_rate(i, 10)
            logs.append(da.step([src[i % 2]], {k: v.detach() for k, v in trg[i % 2].items()}))
            da.update_ema()
        torch.cuda.synchronize()
        assert da.model_select.current == branch
        # the deterministic half: torch raises on any operation of its own that synchronises (.item(), .cpu(), a blocking copy)
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")  # ("a prototype feature")
            torch.cuda.set_sync_debug_mode("error")
        try:
            da.adjust_learning_rate(3, 10)
            da.step([src[1]], {k: v.detach() for k, v in trg[1].items()})
            da.update_ema()
        finally:
            torch.cuda.set_sync_debug_mode("default")
        torch.cuda.synchronize()
        # the wall-clock half (covers the C library's side as well)
        big = torch.randn(8192, 8192, device=DEV)
        for _ in range(3):
            big = big @ big * 1e-4
        tries = []
        for attempt in range(3):  # a wall-clock property on a shared box: one clean run out of three proves it (a step that
            torch.cuda.synchronize()  # reads something back from the device fails every time)
            t0 = time.perf_counter()
            for _ in range(40):
                big = big @ big * 1e-4
            queued = time.perf_counter() - t0
            da.adjust_learning_rate(3 + attempt, 10)
            log = da.step([src[(1 + attempt) % 2]], {k: v.detach() for k, v in trg[(1 + attempt) % 2].items()})
            da.update_ema()
            issued = time.perf_counter() - t0
            torch.cuda.synchronize()
            drained = time.perf_counter() - t0
            tries.append((round(queued, 4), round(issued, 4), round(drained, 4)))
            if queued < 0.2 * drained and issued < 0.6 * drained:  # the matmuls really were a backlog, and the step went in behind it
                break
        else:
            raise AssertionError((branch, tries))
        # ... and the log is complete once somebody looks (both sides: "prior dynamic" exists only on the dynamic side)
        keys = set(log.keys())
        assert ("prior dynamic confidence ma" in keys) == (branch == 1), sorted(keys)
        assert {"prior static confidence ma", "prior EMA confidence ma", "model confidence ma", "dev avg prior static"} <= keys
        assert np.isfinite(float(log["Total target loss"])) and da.model_select.current == branch
        del da, model, big
        torch.cuda.empty_cache()


@pytest.mark.parametrize("nshards", [2, 4, 8])
def test_step_sharded_matches_the_oracle_emulation_of_two_ranks(tmp_path, conv_mode, nshards):
    """``step_sharded`` with `nshards` micro-batches (= what that many data-parallel ranks compute: rank-local batch
    statistics, averaged gradients, one switch decision from averaged confidences, summed prototype statistics, averaged
    running statistics) against the oracle's sequential emulation on the CPU (oracle/step.py step_sharded).  8 and 4 are
    the micro-batch counts `bench.py --global-batch 32` runs per rank at N = 1 and N = 2 (BASELINE config 4's shapes)."""
    if conv_mode != "f16x2":
        pytest.skip("default conv mode only")
    from onda_amd.config import hybrid_switch_cfg
    from onda_amd.framework.handlers import get_adapt_method, get_model
    from onda_amd.framework.model import deeplabv2
    from onda_amd.framework.domain_adaptation.methods.adaptation_model import switch_batch_statistics
    from onda_amd.synthetic import fill_state_dict, synth_batch, synth_tensor
    from oracle import model as omodel
    from oracle.step import OracleAdapter
    cfg, spec = hybrid_switch_cfg(128, 64, DEV, str(tmp_path), batch_size=2)
    model = get_model(cfg, 19)
    fill_state_dict(model, 1, 40.0)
    da = get_adapt_method(cfg)(model, cfg, spec)
    proto_src = [synth_batch(2, 64, 128, seed=100 + i) for i in range(2)]
    shards = [({k: v for k, v in synth_batch(2, 64, 128, seed=300 + r).items()}, synth_batch(2, 64, 128, seed=400 + r))
              for r in range(nshards)]
    torch.manual_seed(123)
    masks = [omodel.draw_drop_mask(2) for _ in range(2 + 3 * nshards)]
    it = iter(masks)
    deeplabv2.drop_mask_fn = lambda B, C, p, dev: next(it).to(dev)
    try:
        da.update_dynamic()
        switch_batch_statistics(da.model, False)
        da.calculate_prototypes(proto_src, save=False)
        switch_batch_statistics(da.model, True)
        da.optimizer.zero_grad()
        before = {k: v.detach().float().cpu().clone() for k, v in da.model.state_dict().items() if v.is_floating_point() and v.dim() > 0}
        da.adjust_learning_rate(0, 6)
        log = da.step_sharded([([s], t) for s, t in shards])
    finally:
        deeplabv2.drop_mask_fn = deeplabv2._default_drop_mask
    # oracle: same state, same masks (prototype batches consumed masks 0,1; then per rank: source, target student, teacher)
    sd = {k: synth_tensor(k, torch.empty(shape, dtype=dt), 1, 40.0).to(dt) for k, shape, dt in omodel.state_spec()}
    from onda_amd.synthetic import synth_prototypes
    ad = OracleAdapter(sd, synth_prototypes())
    ad.refresh_dynamic()
    torch.manual_seed(123)
    ad.proto = ad.initial_prototypes(proto_src)  # draws the same two masks
    rank_masks = [tuple(masks[2 + 3 * r: 5 + 3 * r]) for r in range(nshards)]
    ref = ad.step_sharded(shards, rank_masks)
    assert ad.switch.current == da.model_select.current
    for key in ("buff_loss", "Total target loss", "ce_loss", "rce_loss", "regularization_loss", "pseudolabel_pixel_num",
                "output & prototype agreement", "mean_prototype_intensity_values", "model confidence ma",
                "prior static confidence ma", "prior EMA confidence ma", "prior confidence ma", "prototypes confidence ma",
                "pseudolabel confidence confidence ma"):
        mine, want = log[key], ref[key]
        mine = mine.item() if isinstance(mine, torch.Tensor) else float(mine)
        want = want.item() if isinstance(want, torch.Tensor) else float(want)
        assert _log_close(mine, want, key, 0, 306), (key, mine, want)
    np.testing.assert_allclose(da.prototypes.prototypes.cpu().numpy(), ad.proto[0].numpy(), rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(da.prototypes.squared_mean.cpu().numpy(), ad.proto[1].numpy(), rtol=1e-3, atol=1e-4)
    num = den = 0.0
    for k, b0 in before.items():
        mine, want = da.model.state_dict()[k].detach().float().cpu(), ad.student[k].float()
        num += float(((mine - want) ** 2).sum())
        den += float(((want - b0) ** 2).sum())
    # post-step weights (running statistics included), as updates.  This is HIP against torch-CPU fp32: the floor is the
    # reference's own fp32 noise through a train-mode pass (its step-0 update moves by 0.3 % with its thread count); the
    # same-kernel comparison of the two layouts is test_multirank_gpu.py (1e-4)
    print(f"step_sharded ({nshards} shards) vs oracle emulation, update rel-L2:", (num / den) ** 0.5)
    assert (num / den) ** 0.5 <= 5e-3, (num / den) ** 0.5  # (measured: 1.2e-3)


def test_eval_forward_1024x2048_golden(golden, conv_mode):
    """One frame at the resolution of BASELINE config 5: class map against the reference's (fixture G11)."""
    from onda_amd import ops
    from onda_amd.synthetic import synth_batch
    g = golden("g11_eval_1024x2048")
    m = build_model(1, 3.0).eval()
    x = synth_batch(1, 1024, 2048, seed=11)["image"].to(DEV)
    with torch.no_grad():
        _, o = m(x)
        cls = ops.upsample_argmax(o["out"], (1024, 2048)).cpu().numpy()
    assert o["out"].shape == (1, 19, 129, 257)
    grid = o["out"][:, :, ::4, ::4].cpu().numpy()
    assert np.abs(grid - g["out_grid"]).max() <= 1e-3 * float(g["out_absmax"])
    tie = np.unpackbits(g["tie_mask"])[: cls.size].reshape(cls.shape).astype(bool)
    diff = cls != g["argmax"]
    assert (diff & ~tie).sum() == 0 and diff.mean() < 1e-5, (int((diff & ~tie).sum()), float(diff.mean()))
    np.testing.assert_allclose(digest(o["feat"], 4096)[2:], g["feat_digest"][2:], rtol=0,
                               atol=1e-3 * np.abs(g["feat_digest"][2:]).max())


def test_f16x2_per_layer_against_exact_f32_on_a_pretrained_like_state(conv_mode):
    """What random-init weights and N(0,1) images do not exercise: a state with the dynamic ranges of a trained network --
    per-layer weight scales spread log-uniformly over four decades, BatchNorm gains in [0.05, 4] and offsets of either
    sign, a confident head (so that most per-pixel loss gradients are ~1e-8 next to a few ~1).  One train-mode pass runs
    on the EXACT fp32 MFMA kernels while every convolution's input, weight, output gradient are captured; each layer is
    then re-evaluated by the two-limb f16 kernels on exactly those tensors (so nothing is amplified through the
    network) and held to the exact result: forward, data gradient and weight gradient to 3e-6 relative L2 (two fp32
    accumulations in different orders; each is ~3e-7 from fp64) and 1e-5 of the largest value."""
    if conv_mode != "f16x2":
        pytest.skip("runs once: it switches the conv mode itself")
    from onda_amd import ops
    from onda_amd.framework.model import deeplabv2
    from onda_amd.synthetic import synth_batch
    old_mode = ops.CONV_MODE
    ops.CONV_MODE = "f32"
    try:
        m = build_model(7, 1.0).train()
        g = torch.Generator().manual_seed(77)
        with torch.no_grad():
            for name, p in m.named_parameters():
                if p.dim() == 4:
                    p.mul_(10.0 ** float(torch.empty(1).uniform_(-2, 2, generator=g)))
                elif ".bn" in name or name.startswith("bn1") or "downsample.1" in name:
                    if name.endswith("weight"):
                        p.copy_(torch.exp(torch.empty(p.shape).uniform_(-3.0, 1.4, generator=g)).to(p.device))
                    else:
                        p.copy_((torch.randn(p.shape, generator=g) * 0.5).to(p.device))
            m.layer6.head[1].weight.mul_(3000.0)  # confident head: per-pixel CE gradients from ~1e-8 to ~1
        captured = []

        def hook(mod, inputs, output):
            x, y = inputs[0], output[0]
            rec = {"mod": mod, "x": x.detach(), "w": mod.weight.detach().clone()}
            if y.requires_grad:
                y.register_hook(lambda gr, rec=rec: rec.__setitem__("dy", gr.detach().clone()))
            captured.append(rec)

        handles = [mod.register_forward_hook(hook) for mod in m.modules() if isinstance(mod, deeplabv2.HipConv2d) and mod is not m.conv1]
        b = synth_batch(2, 64, 128, seed=70)
        _, o = m(b["image"].to(DEV))
        ops.seg_losses(o["out"], b["label_res"].to(DEV), 1.0, 0.0, 0.0)[0].backward()
        for h in handles:
            h.remove()
        probs = o["out"].detach().softmax(1).max(1)[0]
        assert probs.median() > 0.99  # the head is confident
        assert len(captured) >= 50
        spans = []
        for rec in captured:
            mod, x, w, dy = rec["mod"], rec["x"], rec["w"], rec.get("dy")
            k, stride, dil, pad = mod.geom()
            pad_to = ops.HEAD_PAD if w.shape[0] == 19 else None
            ref, test = {}, {}
            for mode, dst in (("f32", ref), ("f16x2", test)):
                ops.CONV_MODE = mode
                xi, wi = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
                y, _ = ops.Conv2dFn.apply(xi, wi, None, ops._PackCache(), stride, dil, pad, False, pad_to)
                dst["y"] = y.detach()
                if dy is not None:
                    y.backward(dy)
                    dst["dx"], dst["dw"] = xi.grad, wi.grad
            for key in ref:
                a, r = test[key].double(), ref[key].double()
                scale = r.abs().max()
                if scale == 0:
                    assert a.abs().max() == 0
                    continue
                assert (a - r).norm() <= 3e-6 * r.norm(), (key, tuple(w.shape), float((a - r).norm() / r.norm()))
                assert (a - r).abs().max() <= 1e-5 * scale, (key, tuple(w.shape), float((a - r).abs().max() / scale))
            if dy is not None:
                nz = dy[dy != 0].abs()
                spans.append(float(nz.max() / nz.min()) if nz.numel() else 1.0)
        assert max(spans) > 1e6  # the gradients really span many decades
    finally:
        ops.CONV_MODE = old_mode


def test_eval_after_train_step_uses_fresh_running_statistics():
    """eval forward -> one train-mode forward that moves the running statistics -> eval forward again: the folded
    BatchNorm of the second eval pass must be rebuilt (the kernel writes the buffers behind torch's back)."""
    from onda_amd.framework.domain_adaptation.methods.adaptation_model import switch_batch_statistics
    m = build_model(3, 3.0)
    x = torch.randn(2, 3, 64, 128, device=DEV)
    m.eval()
    with torch.no_grad():
        a = m(x)[1]["out"].clone()
    m.train()
    switch_batch_statistics(m, True)
    with torch.no_grad():
        m(x * 2 + 1)
    m.eval()
    with torch.no_grad():
        b = m(x)[1]["out"].clone()
        fresh = deepcopy(m)
        for mod in fresh.modules():
            mod.__dict__.pop("_fold", None)
            mod.__dict__.pop("_fold_key", None)
        c = fresh(x)[1]["out"]
    assert not torch.equal(a, b)
    assert torch.equal(b, c)


def test_evaluate_path_matches_oracle(tmp_path):
    """da_model.evaluate: forward -> fused upsample/argmax/confusion-matrix on the GPU
    against the oracle's interp -> softmax -> argmax -> np.bincount (BASELINE config 1 flow, small)."""
    from onda_amd.config import hybrid_switch_cfg
    from onda_amd.framework.domain_adaptation.methods.adaptation_model import evaluation
    from onda_amd.framework.utils.func import fast_hist, per_class_iu
    from onda_amd.synthetic import synth_batch, synth_tensor
    from oracle import model as omodel
    cfg, spec = hybrid_switch_cfg(128, 64, DEV, "NONE", batch_size=2)
    m = build_model(1, 3.0)
    ev = evaluation(m, cfg, spec)
    loader = [synth_batch(2, 64, 128, seed=300 + i) for i in range(3)]
    got = ev.evaluate_all({"val": loader})
    sd = {k: synth_tensor(k, torch.empty(shape, dtype=dt), 1, 3.0).to(dt) for k, shape, dt in omodel.state_spec()}
    hist = 0
    with torch.no_grad():
        for b in loader:
            _, o = omodel.forward(b["image"], sd, omodel.BNMode(False))
            _, amap = omodel.upsample_argmax(o["out"], (64, 128))
            for pred, lab in zip(amap, b["label"]):
                hist = hist + fast_hist(lab.numpy().flatten().astype(np.int64), pred.numpy().flatten(), 19)
    ref = per_class_iu(hist)
    assert got["Val mIoU model of val"] == pytest.approx(np.nanmean(ref), rel=1e-6)
    assert got["Val std IoU model of val"] == pytest.approx(np.nanstd(ref), rel=1e-6)
    assert ev.model.training is False  # evaluation.models_default_config keeps eval mode


def test_prediction_utilities_match_oracle(tmp_path):
    """da_model.test_on_samples / run_predictions / save_prediction (reference adaptation_model.py:181-250): the class maps
    of the first samples equal the oracle's interp -> argmax maps, the saved logits are the eval-mode forward's, the
    logged confidence is their mean max-probability."""
    from onda_amd.config import hybrid_switch_cfg
    from onda_amd.framework.domain_adaptation.methods.adaptation_model import evaluation
    from onda_amd.synthetic import synth_batch, synth_tensor
    from oracle import model as omodel
    cfg, spec = hybrid_switch_cfg(128, 64, DEV, "NONE", batch_size=2)
    spec.PREDICTION_SAVE = str(tmp_path / "pred")
    spec.set_ = ("rain", 25)
    ev = evaluation(build_model(1, 3.0), cfg, spec)
    loader = [synth_batch(2, 64, 128, seed=400 + i) for i in range(3)]
    log = ev.test_on_samples({"val": loader}, count=3)
    sd = {k: synth_tensor(k, torch.empty(shape, dtype=dt), 1, 3.0).to(dt) for k, shape, dt in omodel.state_spec()}
    outs = []
    with torch.no_grad():
        for i, b in enumerate(loader):
            _, o = omodel.forward(b["image"], sd, omodel.BNMode(False))
            outs.append(o["out"])
            _, amap = omodel.upsample_argmax(o["out"][:1], (64, 128))
            entry = log[f"Condition val sample {i}"]
            mism = (torch.from_numpy(entry["prediction"]).long() != amap[0].long()).float().mean().item()
            assert mism <= 2e-3, mism  # (ties of the random-weight logits; G1 pins the bit-exact case)
            assert entry["caption"] == "Sample from val" and entry["label"].shape == (64, 128)
    logged = []
    ev.run_predictions(loader, log_fn=logged.append)
    files = sorted((tmp_path / "pred").glob("*/batch-*.pt"))
    assert [f.name for f in files] == ["batch-0.pt", "batch-1.pt", "batch-2.pt"] and files[0].parent.name == "_".join(str(spec.set_))
    for i, f in enumerate(files):
        saved = torch.load(f)
        err = (saved.double() - outs[i].double()).abs().max().item()
        assert err <= 2e-3 * outs[i].abs().max().item(), ("saved logits", err)
        conf = outs[i].softmax(dim=1).max(dim=1)[0].mean().item()
        assert float(logged[i]["Prediction confidence"]) == pytest.approx(conf, rel=2e-3)
        assert logged[i]["Progress"] == pytest.approx(i * 100.0 / 3)
    assert ev.model.training is False


def test_segmentation_step_config2(golden):
    """BASELINE config 2 (segmentation.py:62-88): train-mode forward -> bilinear upsample to the
    label resolution -> CE -> backward -> SGD step, against the oracle on the CPU."""
    import torch.nn.functional as F
    from onda_amd import ops
    from onda_amd.framework.model import deeplabv2
    from onda_amd.optim import ReplaySGD
    from onda_amd.synthetic import synth_batch, synth_tensor
    from oracle import losses, model as omodel
    g = golden("g2_train_small")
    mask = torch.from_numpy(g["drop_mask"])
    b = synth_batch(2, 64, 128, seed=7)
    m = build_model(1, 3.0).train()
    opt = ReplaySGD(m.optim_parameters(2.5e-4), lr=2.5e-4, momentum=0.9, weight_decay=5e-4)
    deeplabv2.drop_mask_fn = lambda B, C, p, dev: mask.to(dev)
    try:
        _, pred = m(b["image"].to(DEV))
    finally:
        deeplabv2.drop_mask_fn = deeplabv2._default_drop_mask
    with torch.no_grad():
        up = ops.UpsampleFn.apply(pred["out"], (64, 128))
    loss = ops.upsample_ce(pred["out"], b["label"].to(DEV))
    # a batch without a single kept pixel: mean over nothing = NaN, as the reference
    assert torch.isnan(ops.upsample_ce(pred["out"].detach(), torch.full_like(b["label"], 255).to(DEV)))
    loss.backward()
    # oracle
    sd = {k: synth_tensor(k, torch.empty(shape, dtype=dt), 1, 3.0).to(dt) for k, shape, dt in omodel.state_spec()}
    names = ["layer6.head.1.weight", "layer6.bottleneck.2.weight", "layer6.bottleneck.2.bias", "layer6.bottleneck.1.bias"]
    for k in names:
        sd[k].requires_grad_(True)
    _, o = omodel.forward(b["image"], sd, omodel.BNMode(True, True, 0.1), mask)
    ref_up = F.interpolate(o["out"], size=(64, 128), mode="bilinear", align_corners=True)
    ref_loss = losses.ce_hard(ref_up, b["label"])
    grads = torch.autograd.grad(ref_loss, [sd[k] for k in names])
    assert loss.item() == pytest.approx(ref_loss.item(), rel=1e-4)
    assert (up.detach().cpu() - ref_up.detach()).abs().max() <= 1e-3 * ref_up.abs().max()
    params = dict(m.named_parameters())
    for k, gr in zip(names, grads):
        assert (params[k].grad.cpu() - gr).abs().max() <= 5e-3 * gr.abs().max(), k
    before = params["layer6.head.1.weight"].detach().clone()
    opt.step()
    moved = (params["layer6.head.1.weight"].detach() - before).abs().max().item()
    assert moved > 0 and all(torch.isfinite(p).all() for p in m.parameters())


def test_segmentation_step_config2_full_size(conv_mode):
    """BASELINE config 2 at its own size -- 512x1024, batch 4 -- through ``SegmentationTrainer.step``: (i) the upsample ->
    cross-entropy head (forward AND backward into the logits) of two of the four images against the oracle on the CPU,
    evaluated on the very logits the HIP model produced; (ii) the whole step at batch 2 (the oracle's train-mode forward /
    backward of the network on the CPU: ~5 TFLOP) -- loss, upsampled logits and the well-conditioned gradients; (iii) the
    batch-4 step itself: loss equal to the oracle's on its logits, every parameter finite and moved."""
    if conv_mode != "f16x2":
        pytest.skip("full-size step: default conv mode only")
    import torch.nn.functional as F
    from onda_amd import ops
    from onda_amd.config import hybrid_switch_cfg
    from onda_amd.framework.domain_adaptation.methods.segmentation import SegmentationTrainer
    from onda_amd.framework.model import deeplabv2
    from onda_amd.synthetic import synth_batch, synth_tensor
    from oracle import losses, model as omodel
    H, W = 512, 1024
    cfg, spec = hybrid_switch_cfg(W, H, DEV, "NONE", batch_size=4)
    spec.LEARNING_RATE, spec.POWER, spec.WEIGHT_DECAY = 2.5e-4, 0.9, 5e-4
    b4 = synth_batch(4, H, W, seed=3000)
    # (i) + (iii): the batch-4 step, logits captured on the way
    m = build_model(1, 3.0).train()
    tr = SegmentationTrainer(m, cfg, spec)
    seen = {}
    real_head = ops.upsample_ce

    def _spy(logits, labels):  # the trainer's loss head (interp -> cross-entropy as one fused pass each way)
        seen["out"] = logits
        logits.retain_grad()
        return real_head(logits, labels)
    torch.manual_seed(5)
    mask4 = omodel.draw_drop_mask(4)
    deeplabv2.drop_mask_fn = lambda B, C, p, dev: mask4.to(dev)
    before = {k: v.detach().clone() for k, v in m.named_parameters() if v.requires_grad}
    ops.upsample_ce = _spy
    try:
        loss4 = tr.step({k: v.to(DEV) for k, v in b4.items()}, 1000)
    finally:
        ops.upsample_ce = real_head
        deeplabv2.drop_mask_fn = deeplabv2._default_drop_mask
    out = seen["out"]
    assert out.shape == (4, 19, 65, 129)
    o_cpu = out.detach().cpu().clone().requires_grad_(True)
    ref_up = F.interpolate(o_cpu, size=(H, W), mode="bilinear", align_corners=True)
    ref_loss = losses.ce_hard(ref_up, b4["label"])
    assert loss4.item() == pytest.approx(ref_loss.item(), rel=1e-5)
    # the head alone on two images: forward value and the gradient that reaches the logits
    two = out.detach()[:2].clone().requires_grad_(True)
    l2 = ops.upsample_ce(two, b4["label"][:2].to(DEV))
    l2.backward()
    # ... and the same head as two passes through the materialised upsampled tensor (the evaluation-side kernels)
    two_b = out.detach()[:2].clone().requires_grad_(True)
    l2b = ops.seg_losses(ops.UpsampleFn.apply(two_b, (H, W)), b4["label"][:2].to(DEV), 1.0, 0.0, 0.0)[0]
    l2b.backward()
    assert l2.item() == pytest.approx(l2b.item(), rel=1e-6)
    assert (two.grad - two_b.grad).abs().max() <= 2e-6 * two_b.grad.abs().max()
    o2 = out.detach()[:2].cpu().clone().requires_grad_(True)
    r2 = losses.ce_hard(F.interpolate(o2, size=(H, W), mode="bilinear", align_corners=True), b4["label"][:2])
    (g2,) = torch.autograd.grad(r2, o2)
    assert l2.item() == pytest.approx(r2.item(), rel=1e-5)
    assert (two.grad.cpu() - g2).abs().max() <= 1e-4 * g2.abs().max()
    moved = [((p.detach() - before[k]).abs().max().item(), k) for k, p in m.named_parameters() if k in before and not k.startswith("layer5.")]
    assert all(torch.isfinite(p).all() for p in m.parameters()) and min(moved)[0] > 0, min(moved)
    del tr, m, out, seen
    torch.cuda.empty_cache()
    # (ii) the whole step at batch 2 against the oracle
    b2 = {k: v[:2] for k, v in b4.items()}
    mask2 = mask4[:2]
    m = build_model(1, 3.0).train()
    deeplabv2.drop_mask_fn = lambda B, C, p, dev: mask2.to(dev)
    try:
        _, pred = m(b2["image"].to(DEV))
    finally:
        deeplabv2.drop_mask_fn = deeplabv2._default_drop_mask
    with torch.no_grad():
        up = ops.UpsampleFn.apply(pred["out"], (H, W))
    loss = ops.upsample_ce(pred["out"], b2["label"].to(DEV))
    loss.backward()
    sd = {k: synth_tensor(k, torch.empty(shape, dtype=dt), 1, 3.0).to(dt) for k, shape, dt in omodel.state_spec()}
    names = ["layer6.head.1.weight", "layer6.bottleneck.2.weight", "layer6.bottleneck.2.bias", "layer6.bottleneck.1.bias"]
    for k in names:
        sd[k].requires_grad_(True)
    _, o = omodel.forward(b2["image"], sd, omodel.BNMode(True, True, 0.1), mask2)
    ref_up = F.interpolate(o["out"], size=(H, W), mode="bilinear", align_corners=True)
    ref_loss = losses.ce_hard(ref_up, b2["label"])
    grads = torch.autograd.grad(ref_loss, [sd[k] for k in names])
    assert loss.item() == pytest.approx(ref_loss.item(), rel=1e-4)
    ref_d = digest(ref_up.detach(), 4096)[2:]
    assert np.abs(digest(up.detach(), 4096)[2:] - ref_d).max() <= 1e-3 * np.abs(ref_d).max()
    params = dict(m.named_parameters())
    for k, gr in zip(names, grads):
        assert (params[k].grad.cpu() - gr).abs().max() <= 5e-3 * gr.abs().max(), k


G8 = {"online_static": ("PROTO_ONLINE", dict(SWITCH_PRIOR_THRESH=1, STATIC_LAMBDA=1, DYNAMIC_LAMBDA=0), 40.0),
      "online_dynamic": ("PROTO_ONLINE", dict(SWITCH_PRIOR_THRESH=0, STATIC_LAMBDA=0, DYNAMIC_LAMBDA=1), 40.0),
      "hswitch": ("PROTO_ONLINE_HSWITCH", dict(SWITCH_PRIOR_THRESH=0.86, SOFT_TRANS=True), 9.0),
      "vswitch": ("PROTO_ONLINE_VSWITCH", dict(SWITCH_PRIOR_THRESH=0.0002, DEV_THRESH=0.0002), 40.0)}


@pytest.mark.parametrize("tag", list(G8))
def test_other_prototype_methods_golden(golden, tmp_path, tag):
    """PROTO_ONLINE (static_model.yml / dynamic_model.yml), HSWITCH and VSWITCH: two steps each
    against the reference's log dict, soft maps and prototypes (fixture G8)."""
    from onda_amd.config import hybrid_switch_cfg
    from onda_amd.framework.handlers import get_adapt_method, get_model
    from onda_amd.framework.model import deeplabv2
    from onda_amd.framework.domain_adaptation.methods.adaptation_model import switch_batch_statistics
    from onda_amd.synthetic import fill_state_dict, synth_batch
    from oracle import model as omodel
    name, over, head_scale = G8[tag]
    g = golden("g8_" + tag)
    cfg, spec = hybrid_switch_cfg(128, 64, DEV, str(tmp_path), batch_size=2)
    cfg.METHOD.ADAPTATION.NAME = name
    spec.pop("GRAY_AREA", None)
    for k, v in over.items():
        spec[k] = v
    model = get_model(cfg, 19)
    fill_state_dict(model, 1, head_scale)
    da = get_adapt_method(cfg)(model, cfg, spec)
    src = [synth_batch(2, 64, 128, seed=100 + i) for i in range(2)]
    trg = [synth_batch(2, 64, 128, seed=200 + i) for i in range(2)]
    torch.manual_seed(123)
    masks = iter([omodel.draw_drop_mask(2) for _ in range(8)])
    deeplabv2.drop_mask_fn = lambda B, C, p, dev: next(masks).to(dev)
    try:
        da.update_dynamic()
        switch_batch_statistics(da.model, False)
        da.calculate_prototypes(src, save=False)
        switch_batch_statistics(da.model, True)
        da.optimizer.zero_grad()
        for s in range(2):
            da.adjust_learning_rate(s, 6)
            log = da.step([src[s]], trg[s])
            da.update_ema()
            assert (trg[s]["stored_predictions"].cpu() - torch.from_numpy(g[f"soft{s}"])).abs().max() < 2e-3
            for k, v in json.loads(str(g[f"log{s}_json"])).items():
                mine = log[k]
                mine = mine.item() if isinstance(mine, torch.Tensor) else float(mine)
                sp = trg[s]["stored_predictions"]
                assert _log_close(mine, v, k, s, sp[0, 0].numel() * sp.shape[0]), (s, k, mine, v)
            np.testing.assert_allclose(da.prototypes.prototypes.cpu().numpy(), g[f"proto{s + 1}"], rtol=1e-3, atol=1e-4)
    finally:
        deeplabv2.drop_mask_fn = deeplabv2._default_drop_mask


@pytest.mark.parametrize("B,H,W", [(1, 75, 141), (3, 33, 65), (2, 97, 50)])
def test_ragged_sizes_against_oracle(B, H, W):
    """Sizes that are not multiples of the stride chain (odd grids, partial 128-row tiles, batch 1 / 3):
    eval forward, the bilinear class map and one train-mode forward+backward against the CPU oracle."""
    from onda_amd import ops
    from onda_amd.synthetic import synth_tensor
    from oracle import losses as olosses, model as omodel
    m = build_model(5, 3.0)
    sd = {k: synth_tensor(k, torch.empty(shape, dtype=dt), 5, 3.0).to(dt) for k, shape, dt in omodel.state_spec()}
    gen = torch.Generator().manual_seed(B * 1000 + H)
    x = torch.randn(B, 3, H, W, generator=gen)
    # eval
    m.eval()
    with torch.no_grad():
        _, o = m(x.to(DEV))
        cls = ops.upsample_argmax(o["out"], (H, W))
        _, ro = omodel.forward(x, sd, omodel.BNMode(False))
        _, rcls = omodel.upsample_argmax(ro["out"], (H, W))
    h, w = ro["out"].shape[2:]
    assert o["out"].shape == (B, 19, h, w) and o["feat"].shape == (B, 256, h, w)
    for key in ("feat", "out"):
        ref = ro[key].numpy()
        assert np.abs(o[key].cpu().numpy() - ref).max() <= 1e-4 * np.abs(ref).max(), key
    diff = (cls.cpu() != rcls)
    if diff.any():  # only numerical ties of the top two classes may differ
        up = torch.nn.functional.interpolate(ro["out"], size=(H, W), mode="bilinear", align_corners=True)
        top2 = up.topk(2, dim=1).values
        margin = (top2[:, 0] - top2[:, 1])[diff]
        assert diff.float().mean() < 1e-4 and margin.max() < 1e-4 * up.abs().max()
    # train-mode forward + backward of CE on the low-resolution logits
    m.train()
    from onda_amd.framework.model import deeplabv2
    mask = omodel.draw_drop_mask(B)
    lab = torch.randint(0, 19, (B, h, w), generator=gen)
    lab[:, 0, :] = 255
    deeplabv2.drop_mask_fn = lambda nb, nc, p, dev: mask.to(dev)
    try:
        _, o = m(x.to(DEV))
        loss = ops.seg_losses(o["out"], lab.to(DEV), 1.0, 0.0, 0.0)[0]
        loss.backward()
    finally:
        deeplabv2.drop_mask_fn = deeplabv2._default_drop_mask
    sdr = {k: v.clone().double().requires_grad_(v.is_floating_point() and "running" not in k) if v.is_floating_point() else v.clone()
           for k, v in sd.items()}
    _, ro = omodel.forward(x.double(), sdr, omodel.BNMode(True, False), mask.double())
    rloss = olosses.ce_hard(ro["out"], lab)
    rloss.backward()
    assert abs(loss.item() - rloss.item()) <= 1e-4 * abs(rloss.item())
    assert np.abs(o["out"].detach().cpu().numpy() - ro["out"].detach().numpy()).max() <= 1e-3 * ro["out"].abs().max().item()
    for name in ("conv1.weight", "layer1.0.conv1.weight", "layer3.2.conv2.weight", "layer6.head.1.weight"):
        g = dict(m.named_parameters())[name].grad.cpu().double()
        r = sdr[name].grad
        assert (g - r).norm() <= 2e-2 * r.norm() + 1e-12, name


def test_train_loop_golden(golden, tmp_path):
    """The OUTER loop the driver calls (train_ouda.py:227-261): `update_cfg_spec` + `hybrid_proDA.train(src, trg, val,
    log_fn=...)` over two synthetic domains against the reference's captured run (fixture G14): initial prototypes +
    evaluation, 3 + 3 steps, replay-buffer additions, `evaluate_update_dynamic` refreshing the dynamic model inside the second
    domain (AUTO_DYNAMIC), epoch-end evaluation + sample maps, the checkpoints and prototype files it leaves behind, and
    the switch moving to the dynamic side -- decided on the device in the default conv mode, on the host in the other."""
    import pickle
    from g14_common import G14, N_MASKS, compare_logs, loaders
    from onda_amd.config import hybrid_switch_cfg
    from onda_amd.framework.handlers import get_adapt_method, get_model
    from onda_amd.framework.model import deeplabv2
    from onda_amd.synthetic import fill_state_dict
    from oracle import model as omodel
    g = golden("g14_train_loop")
    cfg, spec = hybrid_switch_cfg(128, 64, DEV, str(tmp_path), batch_size=2)
    spec.AVG_MONITOR_SIZE, spec.EPOCHS, spec.LEARNING_RATE = G14["monitor"], 1, G14["lr"]
    cfg.TRAINING.PERC_FILL_PER_DOMAIN = G14["perc_fill"]
    model = get_model(cfg, 19)
    fill_state_dict(model, 1, G14["head_scale"])
    init = {k: digest(v.float(), 64)[2:] for k, v in model.state_dict().items()}
    da = get_adapt_method(cfg)(model, cfg, spec)
    src, domains, val = loaders()
    torch.manual_seed(123)
    np.random.seed(G14["np_seed"])
    masks = iter([omodel.draw_drop_mask(2) for _ in range(N_MASKS)])  # the reference run's draws, in its order
    deeplabv2.drop_mask_fn = lambda B, C, p, dev: next(masks).to(dev)
    logs, branch = [], []
    step0 = da.step

    def step(*a, **k):
        out = step0(*a, **k)
        branch.append([da.model_select.current, da.model_select.current_dev])
        return out
    da.step = step
    try:
        first = True
        for d, (set_, loader) in enumerate(zip(((25,), (50,)), domains)):
            spec.set_ = set_
            if d == 1:
                spec["AUTO_DYNAMIC"] = True
                da.dynamic_update_counter = 499
                np.testing.assert_allclose(da.prototypes.prototypes.cpu().numpy(), g["proto_d0"], rtol=1e-3, atol=1e-4)
            spec.SKIP_CALC = bool(spec.SKIP_CALC) or not first
            first = False
            da.update_cfg_spec(spec)
            # (a log sink reads the dictionary when it gets it, as wandb.log does)
            da.train(src, loader, val, log_fn=lambda entry: logs.append(dict(entry.items())))
            assert sorted(os.listdir(tmp_path)) == list(g[f"files_d{d}"])
            assert da.dynamic_update_counter == int(g[f"dynamic_counter_d{d}"])
        assert next(masks, None) is None  # every mask the reference drew was used, none more
    finally:
        deeplabv2.drop_mask_fn = deeplabv2._default_drop_mask
    worst = {}
    compare_logs(g, logs, image_key=lambda entry: entry["prediction"], worst=worst)
    assert np.array_equal(np.array(branch), g["branch"])
    np.testing.assert_allclose(da.prototypes.prototypes.cpu().numpy(), g["proto_d1"], rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(da.prototypes.squared_mean.cpu().numpy(), g["sqmean_d1"], rtol=1e-3, atol=1e-4)
    with open(os.path.join(tmp_path, "proto_(50,).pickle"), "rb") as f:  # the 3-tuple the next run's LOAD_PROTO reads
        stored = pickle.load(f)
    assert len(stored) == 3 and np.allclose(np.asarray(stored[0].cpu() if torch.is_tensor(stored[0]) else stored[0]), g["proto_d1"], rtol=1e-3, atol=1e-4)
    assert [i for i, _ in src.added] == list(g["added_index"])
    for (_, mine), ref in zip(src.added, g["added_maps"]):
        assert (mine.numpy() != ref).mean() <= 5e-3
    names, dg = list(g["state_names_d1"]), g["state_digest_d1"]
    for who, mod, tol in (("static.", da.static_model, 0.0), ("dynamic.", da.dynamic_model, 2e-2), ("student.", da.model, 2e-2),
                          ("teacher.", da.ema_model, 2e-2)):
        num = den = 0.0
        for k, v in mod.state_dict().items():
            if v.is_floating_point() and v.dim() > 0:
                row = dg[names.index(who + k)][2:]
                num += ((digest(v.float(), 64)[2:] - row) ** 2).sum()
                den += ((row - init[k]) ** 2).sum()
        assert (den > 0) == (who != "static.") and num ** 0.5 <= tol * den ** 0.5, (who, num ** 0.5, den ** 0.5)
    print("g14 worst relative deviations:", {k: round(v, 6) for k, v in sorted(worst.items(), key=lambda kv: -kv[1])[:8]})


@pytest.mark.parametrize("b_src,b_trg", [(2, 2), (2, 3), (3, 2)])
def test_paired_student_pass_matches_the_two_passes(tmp_path, b_src, b_trg):
    """The student's source-replay and target passes as ONE pass over both batches (row groups) against the same step with
    the two passes one after the other (ONDA_PAIR_STUDENT=0: the reference's order): same kernels, same batch statistics per
    group -- logs, prototypes, running statistics and the weight update agree to summation-order noise at step 0 (and the
    second step, on weights that differ in their last bits, to the usual amplification)."""
    from onda_amd import ops
    from onda_amd.config import hybrid_switch_cfg
    from onda_amd.framework.domain_adaptation.methods import prototypes as pmod
    from onda_amd.framework.domain_adaptation.methods.adaptation_model import switch_batch_statistics
    from onda_amd.framework.handlers import get_adapt_method, get_model
    from onda_amd.framework.model import deeplabv2
    from onda_amd.synthetic import fill_state_dict, synth_batch
    from oracle import model as omodel
    if not ops.row_groups_supported():
        pytest.skip("row groups exist in the f16x2 / dma configuration only")

    def run(paired):
        old, pmod.PAIR_STUDENT = pmod.PAIR_STUDENT, paired
        try:
            cfg, spec = hybrid_switch_cfg(256, 128, DEV, str(tmp_path), batch_size=2)
            model = get_model(cfg, 19)
            fill_state_dict(model, 1, 40.0)
            da = get_adapt_method(cfg)(model, cfg, spec)
            src = [synth_batch(b_src, 128, 256, seed=100 + i) for i in range(2)]
            trg = [synth_batch(b_trg, 128, 256, seed=200 + i) for i in range(2)]
            torch.manual_seed(123)
            # (masks by the batch size they are asked for: the two orders of passes ask in the same order)
            deeplabv2.drop_mask_fn = lambda B, C, p, dev: omodel.draw_drop_mask(B).to(dev)
            try:
                da.update_dynamic()
                switch_batch_statistics(da.model, False)
                da.calculate_prototypes(src, save=False)
                switch_batch_statistics(da.model, True)
                da.optimizer.zero_grad()
                states, logs = [{k: v.detach().double().cpu().clone() for k, v in da.model.state_dict().items()}], []
                for s_ in range(2):
                    da.adjust_learning_rate(s_, 6)
                    log = da.step([src[s_]], trg[s_])
                    da.update_ema()
                    logs.append({k: float(v) for k, v in log.items() if not isinstance(v, dict)})
                    states.append({k: v.detach().double().cpu().clone() for k, v in da.model.state_dict().items()})
                return states, logs, da.prototypes.prototypes.cpu().clone()
            finally:
                deeplabv2.drop_mask_fn = deeplabv2._default_drop_mask
        finally:
            pmod.PAIR_STUDENT = old

    sa, la, pa = run(True)
    sb, lb, pb = run(False)
    for s_ in range(2):
        assert la[s_].keys() == lb[s_].keys()
        for k in la[s_]:
            a, b = la[s_][k], lb[s_][k]
            assert a == b or (np.isnan(a) and np.isnan(b)) or abs(a - b) <= (2e-5 if s_ == 0 else 5e-3) * max(abs(b), 1e-3), (s_, k, a, b)
        num = den = 0.0
        for k in sa[0]:
            if sa[0][k].is_floating_point() and sa[0][k].dim() > 0:
                num += float(((sa[s_ + 1][k] - sb[s_ + 1][k]) ** 2).sum())
                den += float(((sb[s_ + 1][k] - sb[s_][k]) ** 2).sum())
        rel = (num / den) ** 0.5
        print("paired vs two passes, step", s_, "update rel-L2", rel)
        # Not summation order alone: the two groups of a paired pass share ONE power-of-two limb scale per tensor (from the
        # larger group's bound), so a group's operands are rounded to their 22 bits on another grid than in its own pass --
        # the size of the f16x2 arithmetic's own noise, which a train-mode backward pass on random weights turns into 1e-4
        # of the update against exact fp32 (test_f16x2_steps_track_the_exact_f32_steps: 1.2e-4 / 4.6e-3) and into 9e-4 here
        assert rel <= (3e-3 if s_ == 0 else 0.6), (s_, rel)
    assert (pa - pb).abs().max() <= 1e-4 * pb.abs().max()
