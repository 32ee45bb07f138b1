"""Launch plumbing of the operator layer: the current stream, profiled launches, NHWC views, conv descriptors, device predicates,
row groups, the stream-K workspace."""
import ctypes
import os
from ctypes import byref

import torch

from .. import _lib
from .._lib import OndaConv, OndaLimbOut, call, query
from . import _state
from ._state import BN_EPS, GN_EPS, GN_GROUPS, HEAD_PAD, STEM_K


def _stream():
    return torch.cuda.current_stream().cuda_stream


_L2_KERNELS = ("conv_l2_kernel<4,2>", "conv_l2_kernel<2,2>", "conv_l2_kernel<4,1>", "conv_l2x_kernel<4,2>", "conv_l2a_kernel")


def _l2_name(M, cout, taps, cin):
    """The device kernel a pre-split forward / data-gradient problem runs on (for bench.py's per-kernel figures)."""
    return _L2_KERNELS[query("onda_conv_l2_kernel_id", M, cout, taps, cin)] if _state.PROFILE is not None else ""


def _launch(name, flops, fn_name, *args, tag=None, issued=None):
    """`issued` (profiling only): a callable returning the share of `flops` the kernel really issues (dead taps / dead
    pixel steps skipped: onda_conv_l2_live_fraction); None = all of them."""
    if _state.PROFILE is None:
        return call(fn_name, *args)
    # a launch under a device predicate does its work only while the flag is set: the flag's value at this point of the
    # stream travels with the entry (profile_entries drops the launches that returned at once -- their flops were not executed)
    live = _state.PREDICATE.clone() if _state.PREDICATE is not None else None
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    call(fn_name, *args)
    e1.record()
    _state.PROFILE.append((name, flops, e0, e1, tag, live, flops * (issued() if issued is not None else 1.0)))


def profile_entries(entries):
    """(name, algorithmic flops, e0, e1, tag, issued flops) of the recorded launches that executed (call after a device
    synchronize).  `issued` <= algorithmic: the kernels skip K-steps that only multiply padding."""
    flags = [e[5] for e in entries if e[5] is not None]
    on = torch.stack([f.reshape(()) for f in flags]).ne(0).tolist() if flags else []
    out, i = [], 0
    for e in entries:
        if e[5] is not None:
            i += 1
            if not on[i - 1]:
                continue
        out.append(e[:5] + (e[6],))
    return out


def _p(t):
    return None if t is None else t.data_ptr()


def _require_cuda(t, what):
    if not t.is_cuda:
        raise RuntimeError(f"onda_amd: {what} must live on the GPU (got {t.device}); the HIP path has no CPU fallback")


def nhwc_ld(t):
    """Row stride of a dense-pixel NHWC view; raises if the view is not pixel-dense."""
    b, h, w, c = t.shape
    ld = t.stride(2)
    if t.stride(3) != 1 or t.stride(1) != w * ld or (b > 1 and t.stride(0) != h * w * ld) or ld < c:
        raise RuntimeError(f"onda_amd: tensor is not a pixel-dense NHWC view: shape {tuple(t.shape)} strides {t.stride()}")
    return ld


def as_nhwc(t):
    """Return `t` (logical [B,H,W,C]) as something the kernels accept, copying only if needed."""
    b, h, w, c = t.shape
    ld = t.stride(2)
    if t.stride(3) == 1 and t.stride(1) == w * ld and (b == 1 or t.stride(0) == h * w * ld) and ld >= c and ld % 4 == 0:
        return t
    return t.contiguous()


def conv_out_size(n, k, stride, dil, pad):
    return (n + 2 * pad - dil * (k - 1) - 1) // stride + 1




class predicated:
    """``with ops.predicated(flag):`` -- no-grad forward passes only (nothing in a backward pass reads the flag)."""

    def __init__(self, flag):
        self.flag = flag

    def __enter__(self):
        self.old, _state.PREDICATE = _state.PREDICATE, self.flag
        return self

    def __exit__(self, *exc):
        _state.PREDICATE = self.old
        return False


def predicates_supported():
    return _state.CONV_MODE == "f16x2" and _state.H2_PATH == "dma"


def _desc(B, Hi, Wi, Cin, Ho, Wo, Cout, k, stride, dil, pad, ldx, ldy, ldr=0, out_os=1, Hf=None, Wf=None, relu=0, split=0):
    return OndaConv(B, Hi, Wi, Cin, Ho, Wo, Cout, k, k, stride, dil, pad, ldx, ldy, ldr, out_os,
                    Ho if Hf is None else Hf, Wo if Wf is None else Wf, int(relu), _p(_state.PREDICATE), int(split), 0)




class row_groups:
    def __init__(self, first_images):
        self.first = int(first_images)

    def __enter__(self):
        self.old, _state.ROW_GROUPS = _state.ROW_GROUPS, self.first
        return self

    def __exit__(self, *exc):
        _state.ROW_GROUPS = self.old
        return False


def row_groups_supported():
    return _state.CONV_MODE == "f16x2" and _state.H2_PATH == "dma" and _state.LIMB_ONLY


def _group_split(B, H, W):
    """GEMM row at which the second row group of a [B,H,W,*] activation starts (0: one group)."""
    return _state.ROW_GROUPS * H * W if 0 < _state.ROW_GROUPS < B else 0


_CONV_WS = {}


def _conv_ws(device):
    """Scratch for the balanced (stream-K) conv schedule: one buffer per (device, stream) --
    launches on one stream are ordered, so consecutive convs can share it."""
    key = (str(device), _stream())
    ws = _CONV_WS.get(key)
    if ws is None:
        ws = _CONV_WS[key] = torch.empty(query("onda_conv_ws_floats"), device=device, dtype=torch.float32)
    return ws


