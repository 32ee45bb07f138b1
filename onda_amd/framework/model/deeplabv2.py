"""Drop-in for the reference's ``framework/model/deeplabv2.py``: DeepLabV2 / ResNet with the
ProDA ASPP head, executed by the gfx950 kernels of ``libonda_hip.so``.

What is kept from the reference: ``get_deeplab_v2(...)`` and its arguments
(deeplabv2.py:442-459); ``forward(x: f32[B,3,H,W]) -> (x1, x2)`` with
``x2 == {"feat": f32[B,256,h,w], "out": f32[B,K,h,w]}`` for ``classifier="ProDA"``
(:375-395); the module tree, so ``state_dict()`` has the same 376 keys, ``modules()`` yields
``nn.BatchNorm2d`` instances whose ``momentum`` / ``track_running_stats`` are read at forward
time, ``deepcopy`` works, and ``optim_parameters(lr)`` walks the tree exactly as :397-439 do
(duplicate entries included); the construction order and init recipe (:210-241, :326-331),
so a given torch seed produces the same initial weights.

What is different: inside ``forward`` activations are NHWC and every op is a HIP kernel;
`feat` / `out` come back as NCHW-shaped views of pixel-major buffers (no copy), which is
also the layout ``prototype_handler.transform`` wants.
"""
import os

import torch
import torch.nn as nn

from onda_amd import ops

affine_par = True


class HipConv2d(nn.Conv2d):
    """nn.Conv2d parameters (OIHW, same state_dict keys) + packed copies for the kernels."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._pack = ops._PackCache()

    def geom(self):
        return self.kernel_size[0], self.stride[0], self.dilation[0], self.padding[0]

    def forward(self, x, want_stats=False, cout_pad=None):
        k, stride, dil, pad = self.geom()
        return ops.Conv2dFn.apply(x, self.weight, self.bias, self._pack, stride, dil, pad, want_stats, cout_pad)


class HipBatchNorm2d(nn.BatchNorm2d):
    """nn.BatchNorm2d state; the arithmetic happens in conv_bn()."""

    def folded(self):
        key = tuple((t.data_ptr(), t._version) for t in (self.weight, self.bias, self.running_mean, self.running_var))
        if getattr(self, "_fold_key", None) != key:
            self._fold = ops.bn_eval_fold(self.weight, self.bias, self.running_mean, self.running_var)
            self._fold_key = key
        return self._fold

    def __deepcopy__(self, memo):
        self.__dict__.pop("_bound_cache", None)
        fold = self.__dict__.pop("_fold", None), self.__dict__.pop("_fold_key", None)
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        from copy import deepcopy
        for k, v in self.__dict__.items():
            new.__dict__[k] = deepcopy(v, memo)
        if fold[0] is not None:
            self._fold, self._fold_key = fold
        return new




def conv_bn(conv, bn, x, relu, residual=None):
    """conv -> BatchNorm -> (+residual) -> (ReLU) on an NHWC tensor.

    train mode (student, EMA teacher): batch statistics from the conv epilogue; running
    statistics move only while ``track_running_stats`` (adaptation_model.py:29-36).
    eval mode (static / dynamic models, evaluation): BN folded into the conv epilogue."""
    k, stride, dil, pad = conv.geom()
    if bn.training:
        # "f16x2" with pre-split operands: the BatchNorm output is written as limb planes only (what the next conv, the
        # residual add and the backward mask read); the conv epilogue then also leaves per-channel extrema (4 stat rows)
        limbs = ops.limb_mode(conv.out_channels)
        y, stats = conv(x, want_stats=4 if limbs else True)
        running = (bn.running_mean, bn.running_var, bn.num_batches_tracked) if bn.track_running_stats else None
        if bn.momentum is None:
            raise NotImplementedError("onda_amd: cumulative-average BatchNorm (momentum=None) is not on the hot path")
        fn = ops.BNTrainLimbFn if stats.shape[1] == 4 else ops.BNTrainFn
        return fn.apply(y, stats, bn.weight, bn.bias, residual, relu, running, bn.momentum)
    scale, shift = bn.folded()
    with torch.no_grad():
        wp = conv._pack.get_fwd(conv.weight)
        limb_out = None
        if ops.limb_mode(conv.out_channels) and ops._use_l2(wp, x.shape[3]):
            # the result feeds convolutions (and residual adds) only: written as limb planes by the conv epilogue itself,
            # scaled by an a-priori bound (ops.fold_bounds; cached per weight / statistics version)
            w = conv.weight
            key = (w.data_ptr(), w._version, id(scale), id(shift))
            hit = getattr(bn, "_bound_cache", None)
            if hit is None or hit[0] != key:
                hit = bn._bound_cache = (key, ops.fold_bounds(w, scale, shift), scale, shift)
            limb_out = hit[1]
        y, _, _ = ops.conv_forward(x, wp, k, stride, dil, pad, conv.out_channels, scale=scale, shift=shift, residual=residual,
                                   relu=relu, limb_out=limb_out)
    return y


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, norm_module=HipBatchNorm2d,
                 norm_grad=False):
        super().__init__()
        # Caffe-style: the stride sits on the first 1x1 (reference :22-24)
        self.conv1 = HipConv2d(inplanes, planes, kernel_size=1, stride=stride, bias=False)
        self.bn1 = norm_module(planes, affine=affine_par)
        self.conv2 = HipConv2d(planes, planes, kernel_size=3, stride=1, padding=dilation, bias=False,
                               dilation=dilation)
        self.bn2 = norm_module(planes, affine=affine_par)
        self.conv3 = HipConv2d(planes, planes * 4, kernel_size=1, bias=False)
        self.bn3 = norm_module(planes * 4, affine=affine_par)
        if not norm_grad:
            for bn in (self.bn1, self.bn2, self.bn3):
                for p in bn.parameters():
                    p.requires_grad = False
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        # x feeds conv1 and the shortcut (identity into bn3's residual add, or the downsample conv): their gradients meet
        # in one buffer instead of an elementwise add (ops.GradSink)
        ops.share_grad(x)
        y = conv_bn(self.conv1, self.bn1, x, True)
        y = conv_bn(self.conv2, self.bn2, y, True)
        if self.downsample is not None:
            residual = conv_bn(self.downsample[0], self.downsample[1], x, False)
        else:
            residual = x
        return conv_bn(self.conv3, self.bn3, y, True, residual)


class SEBlock(nn.Module):
    def __init__(self, inplanes, r=16):
        super().__init__()
        self.global_pool = nn.AdaptiveAvgPool2d((1, 1))
        self.se = nn.Sequential(nn.Linear(inplanes, inplanes // r), nn.ReLU(inplace=True),
                                nn.Linear(inplanes // r, inplanes), nn.Sigmoid())

    def forward(self, x):
        return ops.SEScaleFn.apply(x, self.se[0].weight, self.se[0].bias, self.se[2].weight, self.se[2].bias)


_rank_generators = {}


def _default_drop_mask(batch, channels, p, device):
    """What F.dropout2d draws: one Bernoulli(1-p) per (image, channel), scaled by 1/(1-p).  With several ranks every
    rank draws from its own stream (seed + rank offset): the same seed everywhere would drop the same feature channels
    of every rank's different micro-batch."""
    from onda_amd import dist as odist
    gen = None
    if odist.is_on():
        key = str(device)
        gen = _rank_generators.get(key)
        if gen is None:
            gen = _rank_generators[key] = torch.Generator(device=device)
            gen.manual_seed(torch.initial_seed() + odist.seed_offset())
    return torch.empty(batch, channels, 1, 1, device=device).bernoulli_(1 - p, generator=gen).div_(1 - p)


# tests replace this to inject the masks the CPU oracle used
drop_mask_fn = _default_drop_mask

# masks handed to the next forward passes explicitly (the adaptation step draws them in the reference's order and then
# runs the passes in ITS order): consumed first-in first-out by Classifier_Module2.forward
_forced_masks = []


def force_mask(mask):
    _forced_masks.append(mask)


def draw_mask(model, batch, device):
    """The Dropout2d mask the next training-mode forward of `model` would draw (None when dropout is inactive)."""
    head = model.layer6.head[0]
    if not (head.training and head.p > 0):
        return None
    return drop_mask_fn(batch, 256, head.p, device).reshape(batch, 256).contiguous()


class Classifier_Module2(nn.Module):
    """ProDA ASPP head (reference :117-257): 1x1 + four dilated 3x3 branches, each GN+ReLU,
    concat -> SE -> 3x3 -> GN -> Dropout2d = feat -> 1x1 = out."""

    def __init__(self, inplanes, dilation_series, padding_series, num_classes, droprate=0.1, use_se=True):
        super().__init__()
        self.num_classes = num_classes
        self.conv2d_list = nn.ModuleList()
        self.conv2d_list.append(nn.Sequential(
            HipConv2d(inplanes, 256, kernel_size=1, stride=1, padding=0, dilation=1, bias=True),
            nn.GroupNorm(num_groups=32, num_channels=256, affine=True), nn.ReLU(inplace=True)))
        for dilation, padding in zip(dilation_series, padding_series):
            self.conv2d_list.append(nn.Sequential(
                HipConv2d(inplanes, 256, kernel_size=3, stride=1, padding=padding, dilation=dilation, bias=True),
                nn.GroupNorm(num_groups=32, num_channels=256, affine=True), nn.ReLU(inplace=True)))
        mods = [SEBlock(256 * (len(dilation_series) + 1))] if use_se else []
        mods += [HipConv2d(256 * (len(dilation_series) + 1), 256, kernel_size=3, stride=1, padding=1, dilation=1,
                           bias=True),
                 nn.GroupNorm(num_groups=32, num_channels=256, affine=True)]
        self.bottleneck = nn.Sequential(*mods)
        self.head = nn.Sequential(nn.Dropout2d(droprate),
                                  HipConv2d(256, num_classes, kernel_size=1, padding=0, dilation=1, bias=False))
        # init recipe of the reference (:210-241); note that its conv2d_list loop looks at the
        # nn.Sequential wrappers, so it touches nothing there
        for m in self.bottleneck:
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight.data, a=0, mode="fan_in")
                m.bias.data.zero_()
            elif isinstance(m, nn.GroupNorm):
                m.weight.data.fill_(1)
                m.bias.data.zero_()
        for m in self.head:
            if isinstance(m, nn.Conv2d):
                m.weight.data.normal_(0, 0.001)

    def forward(self, x, get_feat=False):
        ys, gammas, betas = [], [], []
        ops.share_grad(x)  # five convs read x: their data gradients accumulate in place (ops.GradSink)
        for seq in self.conv2d_list:
            y, _ = seq[0](x)
            ys.append(y)
            gammas.append(seq[1].weight)
            betas.append(seq[1].bias)
        y = ops.GNConcatFn.apply(True, None, *ys, *gammas, *betas)
        rest = list(self.bottleneck)
        if isinstance(rest[0], SEBlock):
            y = rest[0](y)
            rest = rest[1:]
        y, _ = rest[0](y)
        drop = self.head[0]
        chmul = None
        if drop.training and drop.p > 0:
            if _forced_masks:
                chmul = _forced_masks.pop(0)
            else:
                chmul = drop_mask_fn(y.shape[0], y.shape[3], drop.p, y.device).reshape(y.shape[0], y.shape[3]).contiguous()
        feat = ops.GNConcatFn.apply(False, chmul, y, rest[1].weight, rest[1].bias)
        out_pad, _ = self.head[1](feat, cout_pad=ops.HEAD_PAD)
        out = ops.ClassSliceFn.apply(out_pad, self.num_classes)
        if get_feat:
            return {"feat": feat.permute(0, 3, 1, 2), "out": out}
        return out


class ResNetMulti(nn.Module):
    def __init__(self, block, layers, num_classes, multi_level, classifier_module="normal",
                 norm_module=HipBatchNorm2d, norm_grad=False):
        if classifier_module != "ProDA":
            raise NotImplementedError("onda_amd implements the ProDA classifier head (MODEL.CLASSIFIER: 'ProDA'), the one "
                                      "the hybrid_switch / static_model configs use")
        self.multi_level = multi_level
        self.feat = True
        self.inplanes = 64
        super().__init__()
        self.conv1 = HipConv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = norm_module(64, affine=affine_par)
        if not norm_grad:
            for p in self.bn1.parameters():
                p.requires_grad = False
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1, ceil_mode=True)
        self.layer1 = self._make_layer(block, 64, layers[0], norm_module=norm_module, norm_grad=norm_grad)
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2, norm_module=norm_module, norm_grad=norm_grad)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=1, dilation=2, norm_module=norm_module,
                                       norm_grad=norm_grad)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=1, dilation=4, norm_module=norm_module,
                                       norm_grad=norm_grad)
        if self.multi_level:
            self.layer5 = Classifier_Module2(1024, [6, 12, 18, 24], [6, 12, 18, 24], num_classes)
        self.layer6 = Classifier_Module2(2048, [6, 12, 18, 24], [6, 12, 18, 24], num_classes)
        for m in self.modules():  # reference :326-331 (this also overrides the head's own init)
            if isinstance(m, nn.Conv2d):
                m.weight.data.normal_(0, 0.01)
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()

    def _make_layer(self, block, planes, blocks, stride=1, dilation=1, norm_module=HipBatchNorm2d, norm_grad=False):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion or dilation in (2, 4):
            downsample = nn.Sequential(
                HipConv2d(self.inplanes, planes * block.expansion, kernel_size=1, stride=stride, bias=False),
                norm_module(planes * block.expansion, affine=affine_par))
            if not norm_grad:
                for p in downsample[1].parameters():
                    p.requires_grad = False
        layers = [block(self.inplanes, planes, stride, dilation=dilation, downsample=downsample)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes, dilation=dilation))
        return nn.Sequential(*layers)

    def __deepcopy__(self, memo):
        packer = self.__dict__.pop("_packer", None)  # buffers of the packed weights belong to one model instance
        sync = self.__dict__.pop("_onda_grad_sync", None)  # ... and so does its gradient exchange (onda_amd/dist.py)
        try:
            cls = self.__class__
            new = cls.__new__(cls)
            memo[id(self)] = new
            from copy import deepcopy
            for k, v in self.__dict__.items():
                new.__dict__[k] = deepcopy(v, memo)
        finally:
            if packer is not None:
                self.__dict__["_packer"] = packer
            if sync is not None:
                self.__dict__["_onda_grad_sync"] = sync
        return new

    def _stem(self, x):
        bn = self.bn1
        if bn.training:
            return self._stem_train(x, self._running(bn))
        return ops.MaxPoolFn.apply(ops.stem_eval(x, self.conv1.weight, self.conv1._pack, *bn.folded()), True)

    @staticmethod
    def _running(bn):
        return (bn.running_mean, bn.running_var, bn.num_batches_tracked) if bn.track_running_stats else None

    def _stem_train(self, x, running):
        bn = self.bn1
        y, stats = ops.StemConvFn.apply(x, self.conv1.weight, self.conv1._pack, True)
        y = ops.BNTrainFn.apply(y, stats, bn.weight, bn.bias, None, True, running, bn.momentum)
        return ops.MaxPoolFn.apply(y, True)

    def forward(self, x):
        if x.dim() != 4 or x.shape[1] != 3:
            raise ValueError(f"expected an image batch f32[B,3,H,W], got {tuple(x.shape)}")
        ops._require_cuda(x, "image batch")
        x = x.to(torch.float32)
        packer = self.__dict__.get("_packer")
        if packer is None:
            plain = [m for name, m in self.named_modules() if isinstance(m, HipConv2d) and m is not self.conv1
                     and not name.endswith("head.1") and (self.multi_level or not name.startswith("layer5."))]
            packer = self.__dict__["_packer"] = ops.ModelPacker(plain)
        packer.refresh(need_dgrad=torch.is_grad_enabled() and self.training)
        y = self._stem(x)
        y = self.layer1(y)
        y = self.layer2(y)
        y = self.layer3(y)
        x1 = self.layer5(y, True) if self.multi_level else None
        y = self.layer4(y)
        x2 = self.layer6(y, True)
        return x1, x2

    # ---- parameter groups: the same nested walk as the reference (:397-439) -----------------
    def get_1x_lr_params_no_scale(self):
        for top in (self.conv1, self.bn1, self.layer1, self.layer2, self.layer3, self.layer4):
            for mod in top.modules():
                for p in mod.parameters():
                    if p.requires_grad:
                        yield p

    def get_10x_lr_params(self):
        groups = []
        if self.multi_level:
            groups.append(self.layer5.parameters())
        groups.append(self.layer6.parameters())
        for g in groups:
            for p in g:
                yield p

    def optim_parameters(self, lr):
        return [{"params": self.get_1x_lr_params_no_scale(), "lr": lr},
                {"params": self.get_10x_lr_params(), "lr": 10 * lr}]


def get_deeplab_v2(num_classes=19, multi_level=True, layers=[3, 4, 23, 3], classifier="normal",
                   norm_module=HipBatchNorm2d, norm_grad=False):
    if norm_module is nn.BatchNorm2d:
        norm_module = HipBatchNorm2d
    return ResNetMulti(Bottleneck, layers, num_classes, multi_level, classifier_module=classifier,
                       norm_module=norm_module, norm_grad=norm_grad)
