"""BatchNorm (fp32 and limb-row outputs, one or two row groups), folded eval BatchNorm, max-pool, and the ASPP head's GroupNorm +
concat and SE gate."""
import ctypes
import os
from ctypes import byref

import torch

from .. import _lib
from .._lib import OndaConv, OndaLimbOut, call, query
from . import _state
from ._state import BN_EPS, GN_EPS, GN_GROUPS, HEAD_PAD, STEM_K
from .core import _group_split, _p, _stream, as_nhwc, nhwc_ld
from .limbs import Limbs, _stat_tile_rows, amax_slot, is_limb_only, known_amax, limb_mode, limb_only, limbs_of, tag_amax
from .conv import _sink_give, _sink_of, colsum


class BNTrainFn(torch.autograd.Function):
    """Batch-statistics BatchNorm (+residual, +ReLU) on a conv output whose sum / sum-of-squares
    partials came out of the conv epilogue.  Affine parameters are frozen (no dgamma/dbeta).
    Under ``ops.row_groups`` (the stem's BatchNorm in the paired student pass) the two row groups are normalised one
    after the other on their row ranges of the same buffers; only the second group moves the running statistics."""

    @staticmethod
    def _groups(B, H, W):
        split, M = _group_split(B, H, W), B * H * W
        return [(0, M)] if not split else [(0, split), (split, M - split)]

    @staticmethod
    def forward(ctx, y, stats, gamma, beta, residual, relu, running, momentum):
        B, H, W, C = y.shape
        groups = BNTrainFn._groups(B, H, W)
        G = len(groups)
        mean = torch.empty(G, C, device=y.device, dtype=torch.float32)
        invstd = torch.empty_like(mean)
        out = torch.empty_like(y)
        res = as_nhwc(residual) if residual is not None else None
        if res is not None and nhwc_ld(res) != C:
            res = res.contiguous()
        if G > 1 and (nhwc_ld(y) != C or not y.is_contiguous()):
            raise RuntimeError("onda_amd: row groups need a dense conv output")
        amax = amax_slot(y.device) if _state.CONV_MODE == "f16x2" else None  # the output feeds a conv: leave max|out| behind
        for g, (r0, rows) in enumerate(groups):
            part, nrows = stats, stats.shape[0]
            if G > 1:
                # the conv's partial rows cover 256 (128) GEMM rows each: a group takes its own rows when the boundary falls
                # between two of them (every power-of-two image size), one reduction pass over its rows of y otherwise
                bm = _stat_tile_rows(stats)
                if groups[1][0] % bm == 0 and stats.shape[1] == 2:
                    ts = groups[1][0] // bm
                    part = stats[:ts] if g == 0 else stats[ts:]
                    nrows = part.shape[0]
                else:
                    part = torch.empty(query("onda_bn_bwd_ws", rows, C), device=y.device, dtype=torch.float32)
                    n = ctypes.c_int(0)
                    call("onda_bn_stats", y.data_ptr() + 4 * r0 * C, rows, C, C, _p(part), byref(n), _stream())
                    nrows = n.value
            run = running if (running is not None and g == G - 1) else None
            rm, rv, nbt = run if run is not None else (None, None, None)
            call("onda_bn_finalize", _p(part), nrows, C, rows, BN_EPS, _p(mean[g]), _p(invstd[g]), _p(rm), _p(rv), _p(nbt),
                 float(momentum), _stream())
            off = 4 * r0 * C
            call("onda_bn_apply", y.data_ptr() + off, _p(mean[g]), _p(invstd[g]), _p(gamma), _p(beta),
                 res.data_ptr() + off if res is not None else None, out.data_ptr() + off, rows, C, int(relu), _p(amax), _stream())
        if running is not None:
            # the kernel wrote the running buffers through raw pointers: tell torch, so that everything keyed on
            # their version (HipBatchNorm2d.folded) sees the new statistics
            for t in running:
                torch.autograd.graph.increment_version(t)
        if amax is not None:
            tag_amax(out, amax)
        ctx.save_for_backward(y, out if relu else None, mean, invstd, gamma)
        ctx.relu, ctx.has_res, ctx.groups = relu, residual is not None, groups
        ctx.res_sink = _sink_of(residual) if (residual is not None and ctx.needs_input_grad[4]) else None
        return out

    @staticmethod
    def backward(ctx, dout):
        y, out, mean, invstd, gamma = ctx.saved_tensors
        B, H, W, C = y.shape
        dout = dout.contiguous()
        dx = torch.empty_like(y)
        amax = amax_slot(y.device) if _state.CONV_MODE == "f16x2" else None
        need_res = ctx.has_res and ctx.needs_input_grad[4]
        dres = None
        if need_res:
            dres = torch.empty_like(y) if ctx.relu else dout
        for g, (r0, rows) in enumerate(ctx.groups):
            ws = torch.empty(query("onda_bn_bwd_ws", rows, C), device=y.device, dtype=torch.float32)
            off = 4 * r0 * C
            call("onda_bn_bwd", dout.data_ptr() + off, out.data_ptr() + off if out is not None else None, y.data_ptr() + off,
                 _p(mean[g]), _p(invstd[g]), _p(gamma), dx.data_ptr() + off,
                 dres.data_ptr() + off if (need_res and ctx.relu) else None, _p(ws), rows, C, int(ctx.relu), _p(amax), _stream())
        if amax is not None:
            tag_amax(dx, amax)  # dx is the dy of the conv below: data gradient and weight gradient read it
        if need_res:
            dres = _sink_give(ctx.res_sink, dres, ctx.relu)
        return dx, None, None, None, dres, None, None, None






class BNTrainLimbFn(torch.autograd.Function):
    """BNTrainFn whose output exists as limb planes only ("f16x2" / "dma"): the output of a train-mode BatchNorm
    (+residual, +ReLU) is consumed by convolutions, a later residual add and its own backward mask -- all of which
    read limb planes -- so no fp32 copy is written.  `stats`: the conv epilogue's [tiles][4][C] partials (sum, sum of
    squares, min, max); the extrema bound max|out| before the apply pass (csrc/norm_l2.hip).  Backward: dout is an
    ordinary fp32 tensor, the gradient of the conv output goes out as limb planes again (consumed by the data- and
    weight-gradient kernels only)."""

    @staticmethod
    def forward(ctx, y, stats, gamma, beta, residual, relu, running, momentum):
        B, H, W, C = y.shape
        M = B * H * W
        dev = y.device
        # two row groups (ops.row_groups): statistics [2][C]; only the second group moves the running statistics
        split = _group_split(B, H, W)
        mean = torch.empty((2 if split else 1) * C, device=dev, dtype=torch.float32)
        invstd = torch.empty_like(mean)
        xhat_amax = torch.empty_like(mean)
        rm, rv, nbt = running if running is not None else (None, None, None)
        res = limbs_of(residual) if residual is not None else None
        if res is not None and (res.ld != C or tuple(residual.shape) != (B, H, W, C)):
            raise RuntimeError("onda_amd: residual of a BatchNorm must be a dense [B,H,W,C] activation")
        out_amax = amax_slot(dev)
        tile_rows = _stat_tile_rows(stats) if split else 0
        planes = torch.empty(2, M, C, device=dev, dtype=torch.float16)
        # [out > 0] as one bit per element for the backward passes (they would read 2 bytes of `planes` per element instead)
        mask = torch.empty(M * C // 8, device=dev, dtype=torch.uint8) if relu and any(ctx.needs_input_grad) else None
        if _state.FUSE_BN_FINALIZE and nhwc_ld(y) == C:
            # statistics and apply pass in one launch (csrc/norm_l2.hip, grid_publish / grid_wait)
            call("onda_bn_train_l2", _p(y), _p(stats), stats.shape[0], BN_EPS, _p(mean), _p(invstd), _p(rm), _p(rv), _p(nbt),
                 float(momentum), _p(gamma), _p(beta), _p(res.planes) if res is not None else None,
                 _p(res.amax) if res is not None else None, int(relu), _p(xhat_amax), _p(planes), _p(out_amax), M, C, _p(mask),
                 split, tile_rows, 1, _stream())
        else:
            call("onda_bn_finalize_l2", _p(stats), stats.shape[0], C, M, BN_EPS, _p(mean), _p(invstd), _p(rm), _p(rv), _p(nbt),
                 float(momentum), _p(gamma), _p(beta), _p(res.amax) if res is not None else None, int(relu), _p(xhat_amax),
                 _p(out_amax), split, tile_rows, 1, _p(y) if split else None, nhwc_ld(y) if split else 0, _stream())
            call("onda_bn_apply_l2", _p(y), _p(mean), _p(invstd), _p(gamma), _p(beta), _p(res.planes) if res is not None else None,
                 res.plane if res is not None else 0, _p(res.amax) if res is not None else None, _p(planes), M * C, _p(out_amax),
                 M, C, int(relu), _p(mask), split, _stream())
        if running is not None:
            for t in running:
                torch.autograd.graph.increment_version(t)
        lb = Limbs(planes, out_amax, C, M * C)
        ctx.split = split
        ctx.save_for_backward(y, mean, invstd, gamma, xhat_amax)
        ctx.out_limbs = lb if relu and mask is None else None
        ctx.relu_mask = mask
        ctx.relu, ctx.has_res = relu, residual is not None
        ctx.res_sink = _sink_of(residual) if (residual is not None and ctx.needs_input_grad[4]) else None
        return limb_only((B, H, W, C), dev, lb)

    @staticmethod
    def backward(ctx, dout):
        y, mean, invstd, gamma, xhat_amax = ctx.saved_tensors
        B, H, W, C = y.shape
        M = B * H * W
        dev = y.device
        if is_limb_only(dout):
            raise RuntimeError("onda_amd: the gradient of a BatchNorm output must be an fp32 tensor")
        dout = dout.contiguous()
        ws = torch.empty(query("onda_bn_bwd_l2_ws", M, C), device=dev, dtype=torch.float32)
        planes = torch.empty(2, M, C, device=dev, dtype=torch.float16)
        dx_amax = amax_slot(dev)
        need_res = ctx.has_res and ctx.needs_input_grad[4]
        dres = None
        if need_res:
            dres = torch.empty_like(y) if ctx.relu else dout
        ol = ctx.out_limbs
        call("onda_bn_bwd_l2", _p(dout), _p(ol.planes) if ol is not None else None, ol.plane if ol is not None else 0, _p(y),
             _p(mean), _p(invstd), _p(gamma), _p(xhat_amax), _p(planes), M * C, _p(dx_amax),
             _p(dres) if (need_res and ctx.relu) else None, _p(ws), M, C, int(ctx.relu), _p(ctx.relu_mask), ctx.split,
             int(_state.FUSE_BN_FINALIZE), _stream())
        dx = limb_only((B, H, W, C), dev, Limbs(planes, dx_amax, C, M * C))
        if need_res:
            dres = _sink_give(ctx.res_sink, dres, ctx.relu)
        return dx, None, None, None, dres, None, None, None


def fold_bounds(weight, scale, shift):
    """{max_c |scale_c| * sum_k |w_ck|, max_c |shift_c|} as two device floats: the a-priori bound of an eval-mode
    conv + folded BatchNorm output is max|x| * [0] + [1] (once per weight / statistics version; plain torch plumbing)."""
    with torch.no_grad():
        l1 = weight.detach().abs().sum(dim=(1, 2, 3))
        return torch.stack([(scale.abs() * l1).max(), shift.abs().max()]).float().contiguous()


def bn_eval_fold(gamma, beta, rm, rv):
    C = gamma.numel()
    scale = torch.empty(C, device=gamma.device, dtype=torch.float32)
    shift = torch.empty_like(scale)
    call("onda_bn_fold", _p(gamma), _p(beta), _p(rm), _p(rv), BN_EPS, _p(scale), _p(shift), C, _stream())
    return scale, shift


class MaxPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, limbs=False):
        B, Hi, Wi, C = x.shape
        Ho, Wo = -(-(Hi - 1) // 2) + 1, -(-(Wi - 1) // 2) + 1  # k3 s2 p1 ceil_mode
        if (Ho - 1) * 2 - 1 >= Hi:
            Ho -= 1
        if (Wo - 1) * 2 - 1 >= Wi:
            Wo -= 1
        idx = torch.empty(B, Ho, Wo, C, device=x.device, dtype=torch.uint8)
        x = x.contiguous()
        slot = known_amax(x)
        ctx.save_for_backward(idx)
        ctx.in_shape = (B, Hi, Wi, C)
        if limbs and slot is not None and limb_mode(C):
            # the pooled stem output feeds convolutions only: written as their operand (limb rows scaled by max|x|, which a
            # maximum over windows cannot pass) -- no fp32 copy, no split pass
            planes = torch.empty(2, B * Ho * Wo, C, device=x.device, dtype=torch.float16)
            call("onda_maxpool_fwd_limbs", _p(x), _p(slot), _p(planes), _p(idx), B, Hi, Wi, C, Ho, Wo, _stream())
            return limb_only((B, Ho, Wo, C), x.device, Limbs(planes, slot, C, B * Ho * Wo * C))
        y = torch.empty(B, Ho, Wo, C, device=x.device, dtype=torch.float32)
        call("onda_maxpool_fwd", _p(x), _p(y), _p(idx), B, Hi, Wi, C, Ho, Wo, _stream())
        if slot is not None:  # a max over windows of x >= 0 (behind the stem's ReLU) cannot pass max|x|: no max pass over y
            tag_amax(y, slot)
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        B, Hi, Wi, C = ctx.in_shape
        dy = dy.contiguous()
        dx = torch.empty(B, Hi, Wi, C, device=dy.device, dtype=torch.float32)
        call("onda_maxpool_bwd", _p(dy), _p(idx), _p(dx), B, Hi, Wi, C, dy.shape[1], dy.shape[2], _stream())
        return dx, None


class GNConcatFn(torch.autograd.Function):
    """GroupNorm(32)+ReLU of n conv outputs written side by side into one NHWC buffer
    (the ASPP concat without the torch.cat copy); with n == 1, optional ReLU and an optional
    per-(image, channel) multiplier (the Dropout2d mask) it is the bottleneck GroupNorm."""

    @staticmethod
    def forward(ctx, relu, chmul, *args):
        n = len(args) // 3
        ys, gammas, betas = args[:n], args[n:2 * n], args[2 * n:]
        B, H, W, C = ys[0].shape
        HW = H * W
        cat = torch.empty(B, H, W, C * n, device=ys[0].device, dtype=torch.float32)
        ws = torch.empty(query("onda_gn_ws", B, HW, C), device=cat.device, dtype=torch.float32)
        # the buffer feeds a convolution (through the SE gate, or directly: the class head): the apply passes leave max|cat|
        amax = amax_slot(cat.device) if _state.CONV_MODE == "f16x2" else None
        means, rstds = [], []
        for i in range(n):
            mean = torch.empty(B * GN_GROUPS, device=cat.device, dtype=torch.float32)
            rstd = torch.empty_like(mean)
            sl = cat[..., i * C:(i + 1) * C]
            call("onda_gn_fwd", _p(ys[i]), nhwc_ld(ys[i]), _p(gammas[i]), _p(betas[i]), _p(chmul), _p(sl), C * n,
                 _p(mean), _p(rstd), _p(ws), B, HW, C, GN_GROUPS, GN_EPS, int(relu), _p(amax), _stream())
            means.append(mean)
            rstds.append(rstd)
        if amax is not None:
            tag_amax(cat, amax)
        ctx.save_for_backward(cat, chmul, *ys, *gammas, *means, *rstds)
        ctx.n, ctx.relu = n, relu
        return cat

    @staticmethod
    def backward(ctx, dcat):
        n = ctx.n
        saved = ctx.saved_tensors
        cat, chmul = saved[0], saved[1]
        ys, gammas = saved[2:2 + n], saved[2 + n:2 + 2 * n]
        means, rstds = saved[2 + 2 * n:2 + 3 * n], saved[2 + 3 * n:2 + 4 * n]
        B, H, W, C = ys[0].shape
        HW = H * W
        dcat = as_nhwc(dcat)
        ldd = nhwc_ld(dcat)
        ws = torch.empty(query("onda_gn_ws", B, HW, C), device=cat.device, dtype=torch.float32)
        dys, dgs, dbs = [], [], []
        for i in range(n):
            dx = torch.empty(B, H, W, C, device=cat.device, dtype=torch.float32)
            dg = torch.empty(C, device=cat.device, dtype=torch.float32)
            db = torch.empty_like(dg)
            call("onda_gn_bwd", dcat.data_ptr() + 4 * i * C, ldd, cat.data_ptr() + 4 * i * C, C * n, _p(ys[i]),
                 nhwc_ld(ys[i]), _p(gammas[i]), _p(chmul), _p(means[i]), _p(rstds[i]), _p(dx), _p(dg), _p(db), _p(ws),
                 B, HW, C, GN_GROUPS, int(ctx.relu), _stream())
            dys.append(dx)
            dgs.append(dg)
            dbs.append(db)
        return (None, None, *dys, *dgs, *dbs)




class SEScaleFn(torch.autograd.Function):
    """SEBlock: x * sigmoid(W2 relu(W1 mean_px(x) + b1) + b2)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2):
        B, H, W, C = x.shape
        R = w1.shape[0]
        pooled = colsum(x, alpha=1.0 / (H * W), per_image=True)
        hidden = torch.empty(B, R, device=x.device, dtype=torch.float32)
        gate = torch.empty(B, C, device=x.device, dtype=torch.float32)
        call("onda_se_fc_fwd", _p(pooled), _p(w1), _p(b1), _p(w2), _p(b2), _p(hidden), _p(gate), B, C, R, _stream())
        slot = known_amax(x)
        if slot is not None and limb_mode(C) and x.is_contiguous():
            # the result feeds the bottleneck conv only: written as that conv's operand (limb planes scaled by max|x|, which
            # the sigmoid gate cannot raise) -- no fp32 copy, no max pass, no split pass over the 1280-channel buffer
            planes = torch.empty(2, B * H * W, C, device=x.device, dtype=torch.float16)
            call("onda_chan_scale_limbs", _p(x), _p(gate), _p(planes), B * H * W * C, _p(slot), B, H * W, C, _stream())
            out = limb_only((B, H, W, C), x.device, Limbs(planes, slot, C, B * H * W * C))
        else:
            out = torch.empty_like(x)
            call("onda_chan_scale", _p(x), _p(gate), None, _p(out), B, H * W, C, _stream())
        ctx.save_for_backward(x, pooled, hidden, gate, w1, w2)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, pooled, hidden, gate, w1, w2 = ctx.saved_tensors
        B, H, W, C = x.shape
        R = w1.shape[0]
        dout = dout.contiguous()
        dgate = colsum(dout, x, per_image=True)
        dw1, db1 = torch.empty_like(w1), torch.empty(R, device=x.device, dtype=torch.float32)
        dw2, db2 = torch.empty_like(w2), torch.empty(C, device=x.device, dtype=torch.float32)
        dpooled = torch.empty(B, C, device=x.device, dtype=torch.float32)
        ws = torch.empty(B * R, device=x.device, dtype=torch.float32)
        call("onda_se_fc_bwd", _p(dgate), _p(pooled), _p(hidden), _p(gate), _p(w1), _p(w2), _p(dw1), _p(db1), _p(dw2),
             _p(db2), _p(dpooled), _p(ws), 1.0 / (H * W), B, C, R, _stream())
        dx = torch.empty_like(x)
        call("onda_chan_scale", _p(dout), _p(gate), _p(dpooled), _p(dx), B, H * W, C, _stream())
        return dx, dw1, db1, dw2, db2


