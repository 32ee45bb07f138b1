"""The segmentation head's tail: class slices, fused CE / RCE / MRKLD, softmax statistics, bilinear upsampling (+ fused
cross-entropy / argmax / confusion matrix), the device-side prior select."""
import ctypes
import os
from ctypes import byref

import torch

from .. import _lib
from .._lib import OndaConv, OndaLimbOut, call, query
from . import _state
from ._state import BN_EPS, GN_EPS, GN_GROUPS, HEAD_PAD, STEM_K
from .core import _p, _stream


class ClassSliceFn(torch.autograd.Function):
    """[B,h,w,32] padded head output -> the reference's NCHW `out` f32[B,K,h,w] (a view)."""

    @staticmethod
    def forward(ctx, out_pad, k):
        ctx.k, ctx.pad = k, out_pad.shape[3]
        return out_pad[..., :k].permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, dout):
        B, K, H, W = dout.shape
        g = torch.zeros(B, H, W, ctx.pad, device=dout.device, dtype=torch.float32)
        g[..., :K].copy_(dout.permute(0, 2, 3, 1))
        return g, None


def logits_rows(out):
    """(tensor, ld, N, K) of an NCHW logits tensor laid out pixel-major (as this model
    returns it); copies into a padded pixel-major buffer otherwise."""
    B, K, H, W = out.shape
    ld = out.stride(3)
    if out.stride(1) == 1 and out.stride(2) == W * ld and (B == 1 or out.stride(0) == H * W * ld) and ld >= K:
        return out, ld, B * H * W, K
    buf = torch.zeros(B, H, W, HEAD_PAD, device=out.device, dtype=torch.float32)
    buf[..., :K].copy_(out.detach().permute(0, 2, 3, 1))
    return buf, HEAD_PAD, B * H * W, K


class SegLossFn(torch.autograd.Function):
    """w_ce*CE + w_rce*RCE + w_reg*MRKLD over hard labels, one pass; returns
    (total, ce, rce, mrkld) with gradients flowing through `total` only."""

    @staticmethod
    def forward(ctx, out, labels, w_ce, w_rce, w_reg):
        rows, ld, N, K = logits_rows(out)
        labels = labels.reshape(-1).to(device=out.device, dtype=torch.int64).contiguous()
        result = torch.empty(8, device=out.device, dtype=torch.float32)
        ws = torch.empty(8 * (N // 256 + 1), device=out.device, dtype=torch.float32)
        call("onda_seg_loss_fwd", _p(rows), ld, _p(labels), _p(result), _p(ws), N, K, _stream())
        ctx.save_for_backward(rows, labels, result)
        ctx.meta = (ld, N, K, w_ce, w_rce, w_reg, tuple(out.shape))
        ce, rce, reg = result[0], result[1], result[2]
        total = w_ce * ce + w_rce * rce + w_reg * reg
        ctx.mark_non_differentiable(ce, rce, reg)
        ctx.set_materialize_grads(False)
        return total, ce, rce, reg

    @staticmethod
    def backward(ctx, gtotal, _a, _b, _c):
        if gtotal is None:
            return None, None, None, None, None
        rows, labels, result = ctx.saved_tensors
        ld, N, K, w_ce, w_rce, w_reg, shape = ctx.meta
        B, _, H, W = shape
        dl = torch.empty(B, H, W, ld, device=rows.device, dtype=torch.float32)
        g = gtotal.reshape(1).to(torch.float32).contiguous()
        call("onda_seg_loss_bwd", _p(rows), ld, _p(labels), _p(result), _p(g), w_ce, w_rce, w_reg, _p(dl), N, K,
             _stream())
        return dl[..., :K].permute(0, 3, 1, 2), None, None, None, None


def seg_losses(out, labels, w_ce=1.0, w_rce=0.0, w_reg=0.0):
    return SegLossFn.apply(out, labels, float(w_ce), float(w_rce), float(w_reg))


def softmax_stats(out, want_probs=False, want_argmax=False):
    """Per-pixel softmax of NCHW logits: (mean max-prob 0-dim tensor, probs [N,K] or None, argmax i32[N] or None)."""
    rows, ld, N, K = logits_rows(out)
    probs = torch.empty(N, K, device=out.device, dtype=torch.float32) if want_probs else None
    am = torch.empty(N, device=out.device, dtype=torch.int32) if want_argmax else None
    result = torch.empty(1, device=out.device, dtype=torch.float32)
    ws = torch.empty(N // 256 + 1, device=out.device, dtype=torch.float32)
    call("onda_softmax_stats", _p(rows), ld, _p(probs), K, _p(am), _p(result), _p(ws), N, K, _stream())
    return result[0], probs, am


class UpsampleFn(torch.autograd.Function):
    """nn.Upsample(size, bilinear, align_corners=True) on the pixel-major logits -> NCHW."""

    @staticmethod
    def forward(ctx, out, size):
        rows, ld, _, K = logits_rows(out)
        B, _, h, w = out.shape
        H, W = size
        up = torch.empty(B, K, H, W, device=out.device, dtype=torch.float32)
        call("onda_upsample_fwd", _p(rows), ld, _p(up), B, h, w, K, H, W, _stream())
        ctx.meta = (B, h, w, K, H, W)
        return up

    @staticmethod
    def backward(ctx, dup):
        B, h, w, K, H, W = ctx.meta
        dup = dup.contiguous()
        dl = torch.zeros(B, h, w, HEAD_PAD, device=dup.device, dtype=torch.float32)
        call("onda_upsample_bwd", _p(dup), _p(dl), HEAD_PAD, B, h, w, K, H, W, _stream())
        return dl[..., :K].permute(0, 3, 1, 2), None


class UpsampleCEFn(torch.autograd.Function):
    """loss_calc(interp(out), label): bilinear upsample (align_corners) to the label resolution -> cross-entropy over the
    pixels whose label is not 255, as ONE pass in each direction -- the upsampled logits (and their gradient) exist in
    registers only (csrc/pointwise.hip).
    Label contract: integer class maps with values in [0, K) or the ignore value 255; they travel as uint8, so any value >= K
    (a negative one wraps to >= 128) is IGNORED, as F.cross_entropy(ignore_index=255) ignores 255.  A batch without a single
    kept pixel returns NaN like the reference's mean over zero pixels (utils/loss.py:88-112), and its gradient is all zeros --
    what torch's own nll_loss backward produces for total_weight == 0 -- not NaN: the caller sees the NaN loss."""

    @staticmethod
    def forward(ctx, out, labels):
        rows, ld, _, K = logits_rows(out)
        B, _, h, w = out.shape
        labels = labels.to(device=out.device, dtype=torch.uint8).contiguous()
        H, W = labels.shape[1:]
        result = torch.empty(2, device=out.device, dtype=torch.float32)
        ws = torch.empty(query("onda_upsample_ce_ws", B, H, W), device=out.device, dtype=torch.float32)
        call("onda_upsample_ce_fwd", _p(rows), ld, _p(labels), _p(result), _p(ws), B, h, w, K, H, W, _stream())
        ctx.save_for_backward(rows, labels, result)
        ctx.meta = (ld, B, h, w, K, H, W)
        return result[0]

    @staticmethod
    def backward(ctx, g):
        rows, labels, result = ctx.saved_tensors
        ld, B, h, w, K, H, W = ctx.meta
        dl = torch.empty(B, h, w, ld, device=rows.device, dtype=torch.float32)
        ws = torch.empty(query("onda_upsample_ce_bwd_ws", B, w, K, H), device=rows.device, dtype=torch.float32)
        call("onda_upsample_ce_bwd", _p(rows), ld, _p(labels), _p(result), _p(g.reshape(1).float().contiguous()), 1.0, _p(dl),
             _p(ws), B, h, w, K, H, W, _stream())
        return dl[..., :K].permute(0, 3, 1, 2), None


def upsample_ce(out, labels):
    return UpsampleCEFn.apply(out, labels)


def upsample_argmax(out, size):
    """Fused evaluation tail: class map u8[B,H,W] of interp(out).softmax(1).argmax(1)."""
    rows, ld, _, K = logits_rows(out)
    B, _, h, w = out.shape
    H, W = size
    cls = torch.empty(B, H, W, device=out.device, dtype=torch.uint8)
    call("onda_upsample_argmax", _p(rows), ld, _p(cls), B, h, w, K, H, W, _stream())
    return cls


def upsample_argmax_hist(out, labels, hist, num_classes):
    """Evaluation tail on the GPU: hist[K,K] (int64, accumulated) += confusion matrix of the
    upsampled argmax of `out` against `labels` u8[B,H,W] (values >= K are ignored)."""
    rows, ld, _, K = logits_rows(out)
    B, _, h, w = out.shape
    labels = labels.to(device=out.device, dtype=torch.uint8).contiguous()
    H, W = labels.shape[1:]
    call("onda_upsample_argmax_hist", _p(rows), ld, _p(labels), _p(hist), None, B, h, w, num_classes, H, W, _stream())
    return hist


# ------------------------------------------------------------------------------- device-side switch
def select_prior(flag, a, wa, b, wb):
    """flag ? wb * b : wa * a, elementwise, as a true select (`b` may be garbage when flag == 0)."""
    out = torch.empty_like(a)
    call("onda_select_prior", _p(flag), _p(a.contiguous()), float(wa), _p(b.contiguous()), float(wb), _p(out), a.numel(), _stream())
    return out


def gate_scalar(flag, v):
    """flag ? v : NaN as a device scalar."""
    out = torch.empty(1, device=v.device, dtype=torch.float32)
    call("onda_gate_scalar", _p(flag), _p(v.detach().reshape(1).float()), _p(out), _stream())
    return out[0]


# ------------------------------------------------------------------------------- multi-tensor
