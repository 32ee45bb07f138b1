"""ORACLE (test infrastructure, not product code): CPU restatement of the
reference network forward, DeepLabV2 / ResNet-50 with the ProDA ASPP head.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this package.  The shipped path is the HIP library; it
never routes through here.

Parity status: PINNED by golden vectors produced by importing the reference
itself on CPU (``tests/golden/make_golden.py`` -> G1/G2 fixtures, checked in
``tests/test_oracle_golden.py``).  The reference ships no tests of its own.

The network is written as a pure function over a ``state_dict`` (the 376 keys
of ``framework/model/deeplabv2.py``'s ``ResNetMulti``) in plain fp32 torch ops,
NCHW like the reference.  What each piece follows:

* stem / stages / stage plan   deeplabv2.py:283-325, :375-395
* bottleneck block             deeplabv2.py:53-68 (stride sits on the first 1x1, :22-24;
                               padding = dilation, :29-39)
* ASPP head with SE, GN, Dropout2d, `feat` taken after dropout   deeplabv2.py:117-257
* BatchNorm mode rules         torch.nn.BatchNorm2d as toggled by
                               adaptation_model.py:29-36 and prototypes.py:104-110
"""
from dataclasses import dataclass

import torch
import torch.nn.functional as F

STAGES = (("layer1", 3, 64, 1, 1), ("layer2", 4, 128, 2, 1),
          ("layer3", 6, 256, 1, 2), ("layer4", 3, 512, 1, 4))
ASPP_DILATIONS = (6, 12, 18, 24)
BN_EPS = 1e-5
GN_EPS = 1e-5
GN_GROUPS = 32


@dataclass
class BNMode:
    """How every BatchNorm2d of one forward pass behaves.

    training=True  -> normalise with batch statistics (student and EMA-teacher passes);
                      running stats are updated only if `track` (target pass), momentum 0.1
                      or the per-model override (prototypes.py:55-57).
    training=False -> normalise with the running statistics (static / dynamic passes, eval).
    """
    training: bool = False
    track: bool = True
    momentum: float = 0.1


def _bn(x, sd, prefix, mode: BNMode):
    w, b = sd[prefix + ".weight"], sd[prefix + ".bias"]
    rm, rv = sd[prefix + ".running_mean"], sd[prefix + ".running_var"]
    if mode.training:
        if mode.track:
            sd[prefix + ".num_batches_tracked"] += 1
            return F.batch_norm(x, rm, rv, w, b, True, mode.momentum, BN_EPS)
        return F.batch_norm(x, None, None, w, b, True, mode.momentum, BN_EPS)
    return F.batch_norm(x, rm, rv, w, b, False, mode.momentum, BN_EPS)


def _bottleneck(x, sd, p, stride, dilation, has_down, mode):
    y = F.conv2d(x, sd[p + ".conv1.weight"], stride=stride)
    y = F.relu(_bn(y, sd, p + ".bn1", mode))
    y = F.conv2d(y, sd[p + ".conv2.weight"], padding=dilation, dilation=dilation)
    y = F.relu(_bn(y, sd, p + ".bn2", mode))
    y = F.conv2d(y, sd[p + ".conv3.weight"])
    y = _bn(y, sd, p + ".bn3", mode)
    if has_down:
        r = F.conv2d(x, sd[p + ".downsample.0.weight"], stride=stride)
        r = _bn(r, sd, p + ".downsample.1", mode)
    else:
        r = x
    return F.relu(y + r)


def backbone(x, sd, mode: BNMode):
    y = F.conv2d(x, sd["conv1.weight"], stride=2, padding=3)
    y = F.relu(_bn(y, sd, "bn1", mode))
    y = F.max_pool2d(y, 3, 2, 1, ceil_mode=True)
    for name, blocks, _planes, stride, dil in STAGES:
        for i in range(blocks):
            y = _bottleneck(y, sd, f"{name}.{i}", stride if i == 0 else 1, dil, i == 0, mode)
    return y


def aspp_head(x, sd, drop_mask=None, p="layer6"):
    """Classifier_Module2.forward(get_feat=True).  `drop_mask` is the already
    scaled Dropout2d channel mask f32[B,256,1,1] (None = dropout inactive)."""
    br = []
    for i in range(5):
        q = f"{p}.conv2d_list.{i}"
        if i == 0:
            y = F.conv2d(x, sd[q + ".0.weight"], sd[q + ".0.bias"])
        else:
            d = ASPP_DILATIONS[i - 1]
            y = F.conv2d(x, sd[q + ".0.weight"], sd[q + ".0.bias"], padding=d, dilation=d)
        y = F.relu(F.group_norm(y, GN_GROUPS, sd[q + ".1.weight"], sd[q + ".1.bias"], GN_EPS))
        br.append(y)
    cat = torch.cat(br, 1)
    pooled = cat.mean(dim=(2, 3))
    z = F.relu(F.linear(pooled, sd[p + ".bottleneck.0.se.0.weight"], sd[p + ".bottleneck.0.se.0.bias"]))
    z = torch.sigmoid(F.linear(z, sd[p + ".bottleneck.0.se.2.weight"], sd[p + ".bottleneck.0.se.2.bias"]))
    y = cat * z[:, :, None, None]
    y = F.conv2d(y, sd[p + ".bottleneck.1.weight"], sd[p + ".bottleneck.1.bias"], padding=1)
    y = F.group_norm(y, GN_GROUPS, sd[p + ".bottleneck.2.weight"], sd[p + ".bottleneck.2.bias"], GN_EPS)
    feat = y * drop_mask if drop_mask is not None else y
    out = F.conv2d(feat, sd[p + ".head.1.weight"])
    return {"feat": feat, "out": out}


def forward(x, sd, mode: BNMode = BNMode(), drop_mask=None):
    """ResNetMulti.forward with multi_level=False: returns (None, {"feat","out"})."""
    return None, aspp_head(backbone(x, sd, mode), sd, drop_mask)


def draw_drop_mask(batch, channels=256, p=0.1, generator=None, device="cpu"):
    """The channel mask F.dropout2d(x, p, True) draws ('Randomness'):
    one bernoulli(1-p) per (image, channel) from the default generator, scaled by 1/(1-p)."""
    m = torch.empty(batch, channels, 1, 1, device=device)
    m.bernoulli_(1 - p, generator=generator)
    return m.div_(1 - p)


def upsample_argmax(out, size):
    """Class map of the evaluation path: interp -> softmax -> argmax
    (adaptation_model.py:94-98, :145-153)."""
    up = F.interpolate(out, size=size, mode="bilinear", align_corners=True)
    return up, up.softmax(1).argmax(1)


def state_spec():
    """(key, shape, dtype) of the 376 state_dict entries of ResNetMulti built as
    model_handler.py:16-23 does (ResNet-50, multi_level=True so that the never-executed
    layer5 head is present, classifier "ProDA"), in state_dict order."""
    f32, i64 = torch.float32, torch.int64
    spec = []

    def bn(p, c):
        spec.extend([(p + ".weight", (c,), f32), (p + ".bias", (c,), f32), (p + ".running_mean", (c,), f32),
                     (p + ".running_var", (c,), f32), (p + ".num_batches_tracked", (), i64)])

    spec.append(("conv1.weight", (64, 3, 7, 7), f32))
    bn("bn1", 64)
    inplanes = 64
    for name, blocks, planes, _stride, _dil in STAGES:
        for i in range(blocks):
            p = f"{name}.{i}"
            spec.append((p + ".conv1.weight", (planes, inplanes, 1, 1), f32)); bn(p + ".bn1", planes)
            spec.append((p + ".conv2.weight", (planes, planes, 3, 3), f32)); bn(p + ".bn2", planes)
            spec.append((p + ".conv3.weight", (planes * 4, planes, 1, 1), f32)); bn(p + ".bn3", planes * 4)
            if i == 0:
                spec.append((p + ".downsample.0.weight", (planes * 4, inplanes, 1, 1), f32))
                bn(p + ".downsample.1", planes * 4)
            inplanes = planes * 4
    for p, cin in (("layer5", 1024), ("layer6", 2048)):
        for i in range(5):
            k = 1 if i == 0 else 3
            spec.append((f"{p}.conv2d_list.{i}.0.weight", (256, cin, k, k), f32))
            spec.append((f"{p}.conv2d_list.{i}.0.bias", (256,), f32))
            spec.append((f"{p}.conv2d_list.{i}.1.weight", (256,), f32))
            spec.append((f"{p}.conv2d_list.{i}.1.bias", (256,), f32))
        spec.extend([(f"{p}.bottleneck.0.se.0.weight", (80, 1280), f32), (f"{p}.bottleneck.0.se.0.bias", (80,), f32),
                     (f"{p}.bottleneck.0.se.2.weight", (1280, 80), f32), (f"{p}.bottleneck.0.se.2.bias", (1280,), f32),
                     (f"{p}.bottleneck.1.weight", (256, 1280, 3, 3), f32), (f"{p}.bottleneck.1.bias", (256,), f32),
                     (f"{p}.bottleneck.2.weight", (256,), f32), (f"{p}.bottleneck.2.bias", (256,), f32),
                     (f"{p}.head.1.weight", (19, 256, 1, 1), f32)])
    return spec
