"""Host-side operator layer: torch tensors in, HIP kernels (through the C ABI) underneath.

PyTorch supplies device memory (the caching allocator owns every buffer, workspaces
included), the current HIP stream and the autograd graph; all arithmetic of the hot path
runs in ``libonda_hip.so``.  Activations are dense NHWC tensors ``[B, H, W, C]`` (or channel
slices of one, row stride ``ld``); parameters keep the reference's shapes (OIHW conv weights)
so state_dicts, ``deepcopy`` and optimizers behave exactly as with the reference modules.

Nothing here falls back to eager torch math: a missing library raises at first use.
"""
import ctypes
import os
from ctypes import byref

import torch

from . import _lib
from ._lib import OndaConv, OndaLimbOut, call, query

BN_EPS = 1e-5
GN_EPS = 1e-5
GN_GROUPS = 32
HEAD_PAD = 32  # the 19-class head is computed as a 32-wide GEMM (padded rows are zero)
STEM_K = 160   # 7*7*3 = 147 patch values padded to a multiple of 32


# How the convolutions (forward, data gradient, weight gradient) are evaluated:
#   "f16x2" : fp32 operands scaled by a per-tensor power of two and split into two f16 limbs, three
#             products on the f16 MFMA pipe with fp32 accumulation (the accuracy of an fp32 FMA chain at
#             16/3 of the fp32-MFMA rate; csrc/conv_l2.hip; the arithmetic is described in conv_h2.hip) -- the default;
#   "f32"   : v_mfma_f32_32x32x2_f32 (an exact fp32 fmaf chain; csrc/conv.hip) -- the strict-fp32 leg of bench.py and the
#             yardstick of test_f16x2_steps_track_the_exact_f32_steps; also what a conv whose weight the f16x2 packers do not
#             take (element count not a multiple of 4) runs on.
# (The round-1 "bf16x3" mode -- three bf16 limbs, six products -- was retired in round 4: slower than f16x2, never re-tuned.)
CONV_MODE = os.environ.get("ONDA_CONV_MODE", "f16x2")

# "f16x2": both operands are split into their two limbs BEFOREHAND, as limb rows in HBM (by the producing kernel or one split
# pass); the conv kernels move them to LDS by LDS-DMA only (csrc/conv_l2.hip).  (The round-1 kernels that split the
# activations inside the conv kernel -- ONDA_H2_PATH=reg -- were removed in round 5; the name stays for tools and tests.)
H2_PATH = "dma"

# the multi-GPU gradient exchange installs a callable here: called with the weight Parameter as soon as its gradient of the
# current backward pass has been accumulated in place (autograd's post-accumulate hook fires for it as well, later: the
# exchange counts a parameter once)
GRAD_READY = None

# bench.py sets this to a list to collect (kernel family, algorithmic flops, start event, end event)
# around every conv launch; the events are recorded on the launch stream (torch's current stream)
PROFILE = None


def _stream():
    return torch.cuda.current_stream().cuda_stream


_L2_KERNELS = ("conv_l2_kernel<4,2>", "conv_l2_kernel<2,2>", "conv_l2_kernel<4,1>", "conv_l2x_kernel<4,2>", "conv_l2a_kernel")


def _l2_name(M, cout, taps, cin):
    """The device kernel a pre-split forward / data-gradient problem runs on (for bench.py's per-kernel figures)."""
    return _L2_KERNELS[query("onda_conv_l2_kernel_id", M, cout, taps, cin)] if PROFILE is not None else ""


def _launch(name, flops, fn_name, *args, tag=None, issued=None):
    """`issued` (profiling only): a callable returning the share of `flops` the kernel really issues (dead taps / dead
    pixel steps skipped: onda_conv_l2_live_fraction); None = all of them."""
    if PROFILE is None:
        return call(fn_name, *args)
    # a launch under a device predicate does its work only while the flag is set: the flag's value at this point of the
    # stream travels with the entry (profile_entries drops the launches that returned at once -- their flops were not executed)
    live = PREDICATE.clone() if PREDICATE is not None else None
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    call(fn_name, *args)
    e1.record()
    PROFILE.append((name, flops, e0, e1, tag, live, flops * (issued() if issued is not None else 1.0)))


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


# ------------------------------------------------------------------------------- conv plumbing
# A device int32 the convolutions launched inside ``predicated(flag)`` carry as their predicate: each launch does nothing
# when the flag is 0 (the dynamic model's forward pass under the device-side hybrid switch).  Only the pre-split kernels
# honour it; the other conv modes refuse a predicate (EINVAL) rather than ignore it.
PREDICATE = None


class predicated:
    """``with ops.predicated(flag):`` -- no-grad forward passes only (nothing in a backward pass reads the flag)."""

    def __init__(self, flag):
        self.flag = flag

    def __enter__(self):
        global PREDICATE
        self.old, PREDICATE = PREDICATE, self.flag
        return self

    def __exit__(self, *exc):
        global PREDICATE
        PREDICATE = self.old
        return False


def predicates_supported():
    return CONV_MODE == "f16x2" and H2_PATH == "dma"


def _desc(B, Hi, Wi, Cin, Ho, Wo, Cout, k, stride, dil, pad, ldx, ldy, ldr=0, out_os=1, Hf=None, Wf=None, relu=0, split=0):
    return OndaConv(B, Hi, Wi, Cin, Ho, Wo, Cout, k, k, stride, dil, pad, ldx, ldy, ldr, out_os,
                    Ho if Hf is None else Hf, Wo if Wf is None else Wf, int(relu), _p(PREDICATE), int(split), 0)


# Row groups: inside ``with ops.row_groups(n):`` the first n images of every batch that passes a train-mode BatchNorm are
# one micro-batch and the rest another -- each normalised with its OWN batch statistics, only the second one moving the
# running statistics (the student's source-replay pass under BN_POLICY "freeze" and its target pass, prototypes.py:418-450,
# as ONE pass over both batches: every convolution, every weight gradient and every BatchNorm reduction is launched once
# instead of twice).  Only the pre-split kernels know about groups; the other paths raise.
ROW_GROUPS = 0


class row_groups:
    def __init__(self, first_images):
        self.first = int(first_images)

    def __enter__(self):
        global ROW_GROUPS
        self.old, ROW_GROUPS = ROW_GROUPS, self.first
        return self

    def __exit__(self, *exc):
        global ROW_GROUPS
        ROW_GROUPS = self.old
        return False


def row_groups_supported():
    return CONV_MODE == "f16x2" and H2_PATH == "dma" and LIMB_ONLY


def _group_split(B, H, W):
    """GEMM row at which the second row group of a [B,H,W,*] activation starts (0: one group)."""
    return ROW_GROUPS * H * W if 0 < ROW_GROUPS < B else 0


_CONV_WS = {}


def _conv_ws(device):
    """Scratch for the balanced (stream-K) conv schedule: one buffer per (device, stream) --
    launches on one stream are ordered, so consecutive convs can share it."""
    key = (str(device), _stream())
    ws = _CONV_WS.get(key)
    if ws is None:
        ws = _CONV_WS[key] = torch.empty(query("onda_conv_ws_floats"), device=device, dtype=torch.float32)
    return ws


class H2Weight:
    """Packed weight of the "f16x2" mode: two f16 limb planes of w * 2^e and the device float max|w| that
    defines e (include/onda_hip.h, f16x2 section)."""
    __slots__ = ("limbs", "amax")

    def __init__(self, limbs, amax):
        self.limbs, self.amax = limbs, amax


_AMAX_POOL = {}


AMAX_SLOTS = 2048  # ONDA_AMAX_FLOATS (include/onda_hip.h): 64 slots, one 128-byte line apart
AMAX_POOL_TENSORS = 16384  # tensors served by one zero-filled pool (128 MB)


WEIGHT_POOL_TENSORS = 256  # pools that serve LONG-LIVED scales (packed weights): 2 MB each


def _new_amax_pool(device, tensors=AMAX_POOL_TENSORS):
    buf = torch.zeros(tensors * AMAX_SLOTS, device=device, dtype=torch.float32)
    if buf.is_cuda:  # the zero-fill runs on the creating stream: any other stream that takes slices waits for it once
        done = torch.cuda.Event()
        done.record()
        return [buf, 0, done, {torch.cuda.current_stream().cuda_stream}]
    return [buf, 0, None, None]


def amax_slot(device, long_lived=False):
    """Zeroed device floats for a tensor's running max|x| (slices of a zero-filled pool: one fill kernel per
    16384 tensors instead of one per tensor; a slice is written by exactly one producer and never reused).
    long_lived: scales that outlive a step (the max|w| of packed weights: a frozen static / dynamic model keeps its slices
    for the whole run) come from small pools of their own -- a pool is freed only when every slice of it has died, and one
    long-lived slice used to pin a whole 128 MB activation pool (round-4 advisor)."""
    key = (str(device), bool(long_lived))
    pool = _AMAX_POOL.get(key)
    if pool is None or pool[1] + AMAX_SLOTS > pool[0].numel():
        pool = _AMAX_POOL[key] = _new_amax_pool(device, WEIGHT_POOL_TENSORS if long_lived else AMAX_POOL_TENSORS)
    if pool[2] is not None:
        cur = torch.cuda.current_stream()
        if cur.cuda_stream not in pool[3]:
            cur.wait_event(pool[2])
            pool[3].add(cur.cuda_stream)
    i = pool[1]
    pool[1] = i + AMAX_SLOTS
    return pool[0][i:i + AMAX_SLOTS]


def reserve_amax_slots(device, n):
    """Make sure the next `n` amax_slot() calls are served from a pool that already exists: call on the main stream before
    work is spread over side streams (a refill is safe on any stream -- the others wait for its zero-fill -- but it then
    costs them that wait)."""
    pool = _AMAX_POOL.get((str(device), False))
    if pool is None or pool[1] + n * AMAX_SLOTS > pool[0].numel():
        _AMAX_POOL[(str(device), False)] = _new_amax_pool(device)


def tag_amax(t, slot):
    """Remember that `slot` holds max|t| (valid while t is not modified in place)."""
    try:
        t._onda_scale = (t._version, slot)
    except AttributeError:
        pass
    return t


def known_amax(t):
    hit = getattr(t, "_onda_scale", None)
    return hit[1] if hit is not None and hit[0] == t._version else None


def activation_scale(x):
    """Device float max|x| of an NHWC fp32 activation ("f16x2" mode).  Usually the kernel that produced x
    already left it behind (BatchNorm apply / backward, folded-BN conv epilogue: tag_amax); otherwise one
    reduction pass, shared by every conv that reads the same tensor object."""
    slot = known_amax(x)
    if slot is not None:
        return slot
    B, H, W, C = x.shape
    slot = amax_slot(x.device)
    call("onda_absmax", _p(x), B * H * W, C, nhwc_ld(x), _p(slot), _stream())
    tag_amax(x, slot)
    return slot


class Limbs:
    """An activation as the two f16 limb planes of x * 2^e (include/onda_hip.h, pre-split section):
    planes f16[2, rows, ld], `amax` the device floats that define e."""
    __slots__ = ("planes", "amax", "ld", "plane", "true_amax")

    def __init__(self, planes, amax, ld, plane, true_amax=None):
        self.planes, self.amax, self.ld, self.plane = planes, amax, ld, plane
        # planes whose scale comes from an a-priori BOUND (eval-mode conv outputs) also carry their true max|x|: it is
        # what bounds the next layer's output, so that bounds do not compound
        self.true_amax = true_amax if true_amax is not None else amax


def activation_limbs(x):
    """Limb planes of an NHWC fp32 activation: left behind by its producer, or one split pass shared by
    every conv that reads the same tensor object."""
    hit = getattr(x, "_onda_limbs", None)
    if hit is not None and hit[0] == x._version:
        return hit[1]
    amax = activation_scale(x)
    B, H, W, C = x.shape
    rows = B * H * W
    planes = torch.empty(2, rows, C, device=x.device, dtype=torch.float16)
    call("onda_split_h2", _p(x), rows, C, nhwc_ld(x), _p(planes), C, rows * C, _p(amax), _stream())
    lb = Limbs(planes, amax, C, rows * C)
    try:
        x._onda_limbs = (x._version, lb)
    except AttributeError:
        pass
    return lb


def limb_only(shape, device, limbs):
    """A limb-only activation as autograd sees it: an fp32 [B,H,W,C]-SHAPED tensor without storage behind it (every
    stride 0) that carries the limb planes; gradients with respect to it are ordinary fp32 tensors of that shape."""
    t = torch.empty(1, device=device, dtype=torch.float32).as_strided(tuple(shape), (0,) * len(shape))
    t._onda_limbs = (t._version, limbs)
    return t


def is_limb_only(t):
    return t.dim() == 4 and t.stride() == (0, 0, 0, 0) and t.numel() > 1


def limbs_of(x):
    """Limb planes of an activation: carried by the tensor (its producer wrote them), or one split pass."""
    hit = getattr(x, "_onda_limbs", None)
    if hit is not None and hit[0] == x._version:
        return hit[1]
    if is_limb_only(x):
        raise RuntimeError("onda_amd: a limb-only activation lost its limb planes (it was copied or re-wrapped on the way "
                           "to its consumer); this is a bug in the caller, there is no fp32 copy to fall back to")
    return activation_limbs(x)


def _stat_tile_rows(stats):
    """GEMM rows one row of a conv's statistic partials covers (256 or 128: l2_schedule's tile height, recorded by
    conv_forward).  Two row groups split their partials by it; there is no way to re-derive it from (M, C) alone -- short K
    loops run 128-row tiles (l2_variant_k) -- so a partial table that lost the attribute (detached, cloned, re-wrapped) is an
    error, not a guess (round-5 advisor)."""
    rows = getattr(stats, "_onda_tile_rows", None)
    if not rows:
        raise RuntimeError("onda_amd: BatchNorm with two row groups needs the statistic partials as conv_forward returned "
                           "them (the tensor lost its tile height: it was copied or re-wrapped on the way)")
    return rows


def limb_mode(channels):
    """Do the BatchNorm kernels write their output as limb planes only (no fp32 copy)?"""
    return CONV_MODE == "f16x2" and channels % 32 == 0 and LIMB_ONLY


LIMB_ONLY = True  # (tests flip it to compare with fp32 outputs + split passes)


def materialize(x):
    """fp32 NHWC copy of an activation (diagnostics and tests): limb-only tensors are rebuilt from their planes."""
    if not is_limb_only(x):
        return x
    lb = limbs_of(x)
    B, H, W, C = x.shape
    # limb rows [rows][ld / 32][2][32] -> the two limbs as [rows][ld]
    planes = lb.planes.reshape(B * H * W, lb.ld // 32, 2, 32).permute(2, 0, 1, 3).reshape(2, B * H * W, lb.ld)[:, :, :C].float()
    amax = lb.amax.max()
    e = torch.where((amax > 0) & (amax < 3e38), 15 - torch.frexp(amax)[1], torch.zeros((), dtype=torch.int32, device=x.device))
    return ((planes[0] + planes[1] / query("onda_limb2_scale")) * torch.ldexp(torch.ones((), device=x.device), -e)).reshape(B, H, W, C)


def _use_l2(wp, cin):
    return isinstance(wp, H2Weight) and H2_PATH == "dma" and cin % 32 == 0


def _pack_h2(weight, rows_pad, kp, dgrad, cout_pad):
    cout, cin, kh, kw = weight.shape
    w = weight.detach().contiguous()
    slot = known_amax(weight)  # the forward and the data-gradient packing of one weight version share max|w|
    if slot is None:
        slot = amax_slot(weight.device, long_lived=True)
        call("onda_absmax", _p(w), 1, w.numel(), w.numel(), _p(slot), _stream())  # the whole tensor as one row
        tag_amax(weight, slot)
    dst = torch.empty(2, rows_pad, kp, device=weight.device, dtype=torch.float16)
    call("onda_pack_weight_h2", _p(w), _p(dst), cout, cin, kh * kw, rows_pad, kp, dgrad, cout_pad, _p(slot), _stream())
    return H2Weight(dst, slot)


def pack_weight_fwd(weight, cout_pad=None, kp=None):
    """OIHW -> [Cout_pad][tap*Cin + c] rows of length kp (zero padded); two scaled f16 planes in "f16x2" mode."""
    cout, cin, kh, kw = weight.shape
    taps = kh * kw
    cout_pad = cout_pad or cout
    flat = kp is not None  # a flat, zero-padded K (the stem's patch matrix): consumed as a 1 x 1 conv over kp "channels"
    kp = kp or taps * cin
    # the pre-split kernels walk K in steps of 32 channels of one tap
    if CONV_MODE == "f16x2" and (kp % 32 == 0 if flat else cin % 32 == 0):
        return _pack_h2(weight, cout_pad, kp, 0, cout_pad)
    dst = torch.empty(cout_pad, kp, device=weight.device, dtype=torch.float32)
    call("onda_pack_weight_fwd", _p(weight.detach().contiguous()), _p(dst), cout, cin, taps, cout_pad, kp, _stream())
    return dst


def pack_weight_dgrad(weight, cout_pad=None):
    """OIHW -> [Cin][taps (flipped)][Cout_pad]: the data gradient of a stride-1 conv is a conv of dy with this."""
    cout, cin, kh, kw = weight.shape
    cout_pad = cout_pad or cout
    if CONV_MODE == "f16x2" and cout_pad % 32 == 0:  # (the data gradient's K runs over the output channels)
        return _pack_h2(weight, cin, kh * kw * cout_pad, 1, cout_pad)
    dst = torch.empty(cin, kh * kw, cout_pad, device=weight.device, dtype=torch.float32)
    call("onda_pack_weight_dgrad", _p(weight.detach().contiguous()), _p(dst), cout, cin, kh * kw, cout_pad, _stream())
    return dst


class GradSink:
    """One accumulation buffer for the gradient of an activation with several consumers (a bottleneck block's input:
    first 1x1 conv + identity shortcut or downsample conv; the ASPP input: five convs).  Autograd would add the consumers'
    gradients with one elementwise pass per extra consumer (3 x 4 B per element each); here the first consumer to run
    backward hands ITS gradient tensor to autograd and keeps it as `buf`, every later consumer adds into it in place --
    a conv's data-gradient kernel takes it as the epilogue residual (one extra 4 B read per element) -- and returns None.
    Valid only when EVERY consumer of the tensor follows this protocol (Conv2dFn and the BatchNorm functions' residual
    input): a foreign consumer's gradient would be summed by the engine while later in-place updates are still coming."""
    __slots__ = ("buf", "pending")

    def __init__(self):
        self.buf, self.pending = None, 0


SHARE_GRADS = True  # tests turn this off to compare with autograd's own accumulation


def share_grad(x):
    """Mark `x` (about to be consumed by several of this module's Functions, and by nothing else) for GradSink."""
    if SHARE_GRADS and torch.is_grad_enabled() and x.requires_grad and getattr(x, "_onda_sink", None) is None:
        x._onda_sink = GradSink()  # (marked twice -- an activation read by two modules -- all consumers share one sink)
    return x


def _sink_of(x):
    sink = getattr(x, "_onda_sink", None) if x is not None else None
    if sink is not None:
        sink.pending += 1
    return sink


def _sink_done(sink):
    sink.pending -= 1
    if sink.pending <= 0:
        sink.buf = None  # the engine holds its own reference; a second backward through the graph starts clean


def _sink_give(sink, grad, owned):
    """A consumer's gradient `grad` for the shared tensor (owned: a buffer nobody else references).  Returns what the
    Function hands to autograd."""
    if sink is None:
        return grad
    out = None
    if sink.buf is None:
        out = sink.buf = grad if owned else grad.clone()
    else:
        sink.buf.add_(grad)
    _sink_done(sink)
    return out


def conv_forward(x, wp, k, stride, dil, pad, cout, out=None, scale=None, shift=None, residual=None, relu=False,
                 want_stats=False, limb_out=None):
    """x NHWC view, wp packed [cout][k*k*Cin].  Returns (y, stats partials or None, tiles).
    limb_out: the two device floats of `fold_bounds` -- the result (conv + folded BatchNorm [+ residual] [+ ReLU]) is
    written as limb planes only, scaled by an a-priori bound (include/onda_hip.h, onda_conv2d_fwd_l2_limbs)."""
    _require_cuda(x, "conv input")
    B, Hi, Wi, Cin = x.shape
    ldx = 0 if is_limb_only(x) else nhwc_ld(x)
    Ho, Wo = conv_out_size(Hi, k, stride, dil, pad), conv_out_size(Wi, k, stride, dil, pad)
    if limb_out is not None:
        if out is not None or want_stats or not _use_l2(wp, Cin) or cout % 32 != 0:
            raise RuntimeError("onda_amd: limb-plane conv output needs the pre-split path, a dense result and no statistics")
        dev, M = x.device, B * Ho * Wo
        xl = limbs_of(x)
        res = limbs_of(residual) if residual is not None else None
        if res is not None and tuple(residual.shape) != (B, Ho, Wo, cout):
            raise RuntimeError("onda_amd: residual of a limb-plane conv output must be a dense [B,H,W,C] activation")
        planes = torch.empty(2, M, cout, device=dev, dtype=torch.float16)
        bound, true = amax_slot(dev), amax_slot(dev)
        d = _desc(B, Hi, Wi, Cin, Ho, Wo, cout, k, stride, dil, pad, xl.ld, cout, res.ld if res is not None else 0, relu=relu)
        lo = OndaLimbOut(_p(planes), M * cout, _p(bound), _p(true), _p(limb_out), _p(xl.true_amax),
                         _p(res.planes) if res is not None else None, res.plane if res is not None else 0,
                         _p(res.amax) if res is not None else None, _p(res.true_amax) if res is not None else None)
        _launch(_l2_name(M, cout, k * k, Cin), 2.0 * M * cout * k * k * Cin,
                "onda_conv2d_fwd_l2_limbs", _p(xl.planes), xl.plane, _p(xl.amax), _p(wp.limbs), _p(wp.amax), _p(scale), _p(shift),
                byref(lo), _p(_conv_ws(dev)), byref(d), _stream(), tag=("fwd", M, cout, Cin, k, stride, dil),
                issued=lambda: query("onda_conv_l2_live_fraction", byref(d), 0))
        return limb_only((B, Ho, Wo, cout), dev, Limbs(planes, bound, cout, M * cout, true_amax=true)), None, 0
    if out is None:
        out = torch.empty(B, Ho, Wo, cout, device=x.device, dtype=torch.float32)
    else:  # a caller's buffer is rewritten behind torch's version counter: forget its old max|x| / limb planes
        for attr in ("_onda_scale", "_onda_limbs"):
            if getattr(out, attr, None) is not None:
                delattr(out, attr)
    ldy = nhwc_ld(out)
    ldr = nhwc_ld(residual) if residual is not None else 0
    stats, tiles = None, 0
    l2 = _use_l2(wp, Cin)
    if not l2 and is_limb_only(x):
        raise RuntimeError("onda_amd: a limb-only activation reached a conv that does not take limb planes")
    stats_rows = 4 if (l2 and want_stats == 4) else 2  # 4: + per-channel min / max, for the limb-writing BatchNorm
    split = 0
    if want_stats:
        split = _group_split(B, Ho, Wo)
        if split and not l2:
            raise RuntimeError("onda_amd: row groups (ops.row_groups) need the pre-split conv path")
        tile_rows = ctypes.c_int(0)
        tiles = (query("onda_conv_l2_tiles_m_split", B * Ho * Wo, cout, k * k, Cin, split, 0, byref(tile_rows)) if l2
                 else query("onda_conv_tiles_m", B * Ho * Wo))
        stats = torch.empty(tiles, stats_rows, cout, device=x.device, dtype=torch.float32)
        if l2:  # GEMM rows one partial row covers: depends on the kernel the problem runs on (the BatchNorm row groups need it)
            stats._onda_tile_rows = tile_rows.value
    d = _desc(B, Hi, Wi, Cin, Ho, Wo, cout, k, stride, dil, pad, ldx, ldy, ldr, relu=relu, split=split)
    if l2:
        xl = limbs_of(x)
        d.ldx = xl.ld
        yamax = amax_slot(x.device) if (scale is not None or relu) else None
        _launch(_l2_name(B * Ho * Wo, cout, k * k, Cin), 2.0 * B * Ho * Wo * cout * k * k * Cin,
                "onda_conv2d_fwd_l2", _p(xl.planes), xl.plane, _p(xl.amax), _p(wp.limbs), _p(wp.amax), _p(out), _p(scale),
                _p(shift), _p(residual), _p(stats), stats_rows, _p(_conv_ws(x.device)), _p(yamax), byref(d), _stream(),
                tag=("fwd", B * Ho * Wo, cout, Cin, k, stride, dil),
                issued=lambda: query("onda_conv_l2_live_fraction", byref(d), int(stats is not None)))
        if yamax is not None:
            tag_amax(out, yamax)
        return out, stats, tiles
    if isinstance(wp, H2Weight):
        raise RuntimeError("onda_amd: a pre-split weight reached a conv the pre-split kernels do not take (Cin % 32 != 0)")
    _launch("conv_fwd_kernel<128,%d>" % (128 if cout > 64 else 64), 2.0 * B * Ho * Wo * cout * k * k * Cin, "onda_conv2d_fwd",
            _p(x), _p(wp), _p(out), _p(scale), _p(shift), _p(residual), _p(stats), _p(_conv_ws(x.device)), byref(d), _stream(),
            tag=("fwd", B * Ho * Wo, cout, Cin, k, stride, dil))
    return out, stats, tiles


def conv_dgrad(dy, wpd, k, stride, dil, pad, cin, in_hw, accumulate=None):
    """Data gradient.  dy NHWC [B,Ho,Wo,Cout(_pad)], wpd = pack_weight_dgrad(weight).
    `accumulate`: a dense fp32 [B,Hi,Wi,cin] buffer the gradient is ADDED to (GradSink); returns it."""
    B, Ho, Wo, Co = dy.shape
    Hi, Wi = in_hw
    ldy = 0 if is_limb_only(dy) else nhwc_ld(dy)
    if accumulate is not None:
        fused = (stride == 1 and _use_l2(wpd, Co) and accumulate.is_contiguous() and accumulate.dtype == torch.float32
                 and tuple(accumulate.shape) == (B, Hi, Wi, cin) and not is_limb_only(accumulate))
        if not fused:
            accumulate.add_(conv_dgrad(dy, wpd, k, stride, dil, pad, cin, in_hw))
            return accumulate
        # the epilogue reads the running sum as its residual and stores over it (same thread, same 16 bytes)
        d = _desc(B, Ho, Wo, Co, Hi, Wi, cin, k, 1, dil, dil * (k - 1) - pad, ldy, cin, cin)
        dyl = limbs_of(dy)
        d.ldx = dyl.ld
        _launch(_l2_name(B * Hi * Wi, cin, k * k, Co), 2.0 * B * Ho * Wo * cin * k * k * Co,
                "onda_conv2d_fwd_l2", _p(dyl.planes), dyl.plane, _p(dyl.amax), _p(wpd.limbs), _p(wpd.amax), _p(accumulate), None,
                None, _p(accumulate), None, 2, _p(_conv_ws(dy.device)), None, byref(d), _stream(),
                tag=("dgrad", B * Hi * Wi, cin, Co, k, stride, dil),
                issued=lambda: query("onda_conv_l2_live_fraction", byref(d), 0))
        return accumulate
    if stride == 1:
        dx = torch.empty(B, Hi, Wi, cin, device=dy.device, dtype=torch.float32)
        d = _desc(B, Ho, Wo, Co, Hi, Wi, cin, k, 1, dil, dil * (k - 1) - pad, ldy, cin)
    else:
        if k != 1 or pad != 0:
            raise RuntimeError("onda_amd: strided data gradient is implemented for 1x1 convs only")
        dx = torch.zeros(B, Hi, Wi, cin, device=dy.device, dtype=torch.float32)
        d = _desc(B, Ho, Wo, Co, Ho, Wo, cin, 1, 1, 1, 0, ldy, cin, out_os=stride, Hf=Hi, Wf=Wi)
    if _use_l2(wpd, Co):
        dyl = limbs_of(dy)
        d.ldx = dyl.ld
        Mo = B * Ho * Wo if stride != 1 else B * Hi * Wi
        _launch(_l2_name(Mo, cin, k * k, Co), 2.0 * B * Ho * Wo * cin * k * k * Co,
                "onda_conv2d_fwd_l2", _p(dyl.planes), dyl.plane, _p(dyl.amax), _p(wpd.limbs), _p(wpd.amax), _p(dx), None, None,
                None, None, 2, _p(_conv_ws(dy.device)), None, byref(d), _stream(),
                tag=("dgrad", Mo, cin, Co, k, stride, dil), issued=lambda: query("onda_conv_l2_live_fraction", byref(d), 0))
        return dx
    if is_limb_only(dy):
        raise RuntimeError("onda_amd: a limb-only gradient reached a data-gradient kernel that does not take limb planes")
    if isinstance(wpd, H2Weight):
        raise RuntimeError("onda_amd: a pre-split weight reached a data gradient the pre-split kernels do not take (Cout % 32 != 0)")
    _launch("conv_fwd_kernel<128,%d>" % (128 if cin > 64 else 64), 2.0 * B * Ho * Wo * cin * k * k * Co, "onda_conv2d_fwd",
            _p(dy), _p(wpd), _p(dx), None, None, None, None, _p(_conv_ws(dy.device)), byref(d), _stream(),
            tag=("dgrad", B * Ho * Wo if stride != 1 else B * Hi * Wi, cin, Co, k, stride, dil))
    return dx


def _wgrad_splitk(M, cout, cin, taps, l2=False):
    """Split count over the pixel (K) range: pick the one whose workgroup count best fills whole
    rounds of the resident workgroups (tail effect) net of the slab write+read it costs."""
    if l2:  # pre-split kernel: 256 x 128 tiles, one workgroup per CU; 128 x 128 (two-stage ring), two per CU
        tn = 256 if query("onda_conv_wgrad_l2_variant", cout, cin) == 0 else 128
        tiles = -(-cout // tn) * -(-cin // 128) * taps
        G = query("onda_conv_ws_floats") // (3 * 128 * 128) // (2 if tn == 256 else 1)
        # (the kernel lists a workgroup's K-steps in LDS: at most 2048 steps of 32 pixels per split)
        return max(_best_splitk(M, cout, cin, taps, tiles, G, 3.2e14 if tn == 256 else 2.0e14), -(-M // 65536))
    t = 128 if (cout > 64 and cin > 64) else 64
    tiles = -(-cout // t) * -(-cin // t) * taps
    G = query("onda_conv_ws_floats") // (3 * 128 * 128) * (1 if t == 128 else 2)
    return _best_splitk(M, cout, cin, taps, tiles, G, 1.2e14)


def _best_splitk(M, cout, cin, taps, tiles, G, rate):
    t_ideal = 2.0 * M * cout * cin * taps / rate              # seconds at the kernel's typical rate
    wbytes = 4.0 * cout * cin * taps
    best, best_t = 1, None
    for sk in range(1, 257):
        if sk > 1 and (M // sk < 256 or sk * wbytes > (512 << 20)):
            break
        blocks = tiles * sk
        eff = (blocks / G) / -(-blocks // G)
        est = t_ideal / eff + 2.0 * sk * wbytes / 3e12
        if best_t is None or est < best_t * 0.995:
            best, best_t = sk, est
    return best


# Pixel tables of the pre-split weight gradient (include/onda_hip.h, onda_conv2d_wgrad_l2_table): input pixel of every
# (filter tap, output pixel) of a convolution GEOMETRY, owned here like every other buffer the library works on -- one int32
# tensor per (device, geometry) from torch's allocator, built with one launch at the first backward pass of that geometry
# (eight geometries, 17 MB, in this network), kept for the life of the process; a larger batch builds a larger table and the
# smaller one is dropped when its last launch has been queued (same-stream order; other streams wait for the build's event).
_PIX_TABLES = {}


def _wgrad_pixel_table(d, device):
    """Fill d.pix_table / d.pix_stride for a weight-gradient descriptor (no-op for problems that run without a table)."""
    stride = query("onda_conv2d_wgrad_l2_table_stride", byref(d))
    if stride == 0:
        return
    key = (str(device), d.Hi, d.Wi, d.Ho, d.Wo, d.kh, d.kw, d.stride, d.dil, d.pad)
    hit = _PIX_TABLES.get(key)
    cur = torch.cuda.current_stream(device)
    if hit is None or hit[1] < d.B:
        table = torch.empty(d.kh * d.kw * stride, device=device, dtype=torch.int32)
        call("onda_conv2d_wgrad_l2_table", byref(d), _p(table), cur.cuda_stream)
        done = torch.cuda.Event()
        done.record(cur)
        if hit is not None:
            hit[0].record_stream(cur)  # (launches already queued on this stream still read the smaller table)
        hit = _PIX_TABLES[key] = (table, d.B, stride, done, {cur.cuda_stream})
    elif cur.cuda_stream not in hit[4]:
        cur.wait_event(hit[3])
        hit[4].add(cur.cuda_stream)
    d.pix_table, d.pix_stride = hit[0].data_ptr(), hit[2]


def conv_wgrad(x, dy, k, stride, dil, pad, cout_real, cin_real, flat_k=0, into=None, xscale=None, xlimbs=None):
    """Weight gradient in OIHW.  x NHWC input of the conv, dy NHWC output gradient.  With `into`
    (an existing contiguous gradient tensor) the result is ADDED to it and None is returned.
    xscale: the per-tensor scale of x when the forward pass already computed it ("f16x2" mode)."""
    B, Hi, Wi, Cin = x.shape
    _, Ho, Wo, Co = dy.shape
    taps = k * k
    M = B * Ho * Wo
    l2 = CONV_MODE == "f16x2" and Cin % 32 == 0 and Co % 32 == 0
    sk = _wgrad_splitk(M, Co, Cin, taps, l2)
    slabs = torch.empty(sk, Co, taps, Cin, device=x.device, dtype=torch.float32)
    d = _desc(B, Hi, Wi, Cin, Ho, Wo, Co, k, stride, dil, pad, 0 if is_limb_only(x) else nhwc_ld(x), Co)
    if not l2 and (is_limb_only(x) or is_limb_only(dy)):
        raise RuntimeError("onda_amd: a limb-only tensor reached a weight-gradient kernel that does not take limb planes")
    if l2:
        xl = xlimbs if xlimbs is not None else limbs_of(x)
        dyl = limbs_of(dy)
        d.ldx = xl.ld
        _wgrad_pixel_table(d, x.device)
        _launch("conv_wgrad_l2_kernel<%d>" % query("onda_conv_wgrad_l2_variant", Co, Cin), 2.0 * M * Co * taps * Cin,
                "onda_conv2d_wgrad_l2", _p(xl.planes), xl.plane, _p(xl.amax), _p(dyl.planes), dyl.plane, _p(dyl.amax), _p(slabs),
                dyl.ld, sk, byref(d), _stream(), tag=("wgrad", M, Co, Cin, k, stride, dil, sk),
                issued=lambda: query("onda_conv_wgrad_l2_live_fraction", byref(d), sk))
    else:
        _wgrad_other(x, dy, slabs, sk, d, M, Co, taps, Cin, k, stride, dil)
    return _wgrad_finish(slabs, into, sk, Co, taps, Cin, cout_real, cin_real, flat_k, k, x.device)


def _wgrad_other(x, dy, slabs, sk, d, M, Co, taps, Cin, k, stride, dil):
    _launch("conv_wgrad_kernel<%s>" % ("128,128" if (Co > 64 and Cin > 64) else "64,64"), 2.0 * M * Co * taps * Cin,
            "onda_conv2d_wgrad", _p(x), _p(dy), _p(slabs), nhwc_ld(dy), sk, byref(d), _stream(),
            tag=("wgrad", M, Co, Cin, k, stride, dil, sk))


def _wgrad_finish(slabs, into, sk, Co, taps, Cin, cout_real, cin_real, flat_k, k, device):
    if into is not None:
        call("onda_wgrad_reduce", _p(slabs), _p(into), sk, Co, taps, Cin, cout_real, cin_real, flat_k, 1, _stream())
        return None
    if flat_k:
        dw = torch.empty(cout_real, cin_real, 7, 7, device=device, dtype=torch.float32)
    else:
        dw = torch.empty(cout_real, cin_real, k, k, device=device, dtype=torch.float32)
    call("onda_wgrad_reduce", _p(slabs), _p(dw), sk, Co, taps, Cin, cout_real, cin_real, flat_k, 0, _stream())
    return dw


def _accumulate_target(weight):
    """The parameter's existing .grad if the weight gradient can be added to it in place (second
    backward of a step): saves the separate accumulation pass autograd would run."""
    g = weight.grad
    if g is not None and g.is_contiguous() and g.dtype == torch.float32 and g.shape == weight.shape:
        return g
    return None


def colsum(x, y=None, alpha=1.0, per_image=False):
    """sum over pixels of x (*y): [C] (or [B,C] if per_image).  x, y NHWC views."""
    B, H, W, C = x.shape
    nb, hw = (B, H * W) if per_image else (1, B * H * W)
    ws = torch.empty(query("onda_colsum_ws", nb, hw, C), device=x.device, dtype=torch.float32)
    out = torch.empty(nb, C, device=x.device, dtype=torch.float32)
    call("onda_colsum", _p(x), nhwc_ld(x), _p(y), nhwc_ld(y) if y is not None else 0, _p(out), float(alpha), _p(ws),
         nb, hw, C, _stream())
    return out if per_image else out[0]


# ------------------------------------------------------------------------------- autograd ops
class _PackCache:
    """Packed copies of one conv weight, rebuilt when the parameter changes (version counter or storage)."""

    def __init__(self):
        self.key_f = self.key_d = None
        self.fwd = self.dgrad = None

    @staticmethod
    def _key(w):
        return (w.data_ptr(), w._version, w.device, CONV_MODE)

    def get_fwd(self, w, cout_pad=None, kp=None):
        k = self._key(w)
        if self.key_f != k:
            self.fwd, self.key_f = pack_weight_fwd(w, cout_pad, kp), k
        return self.fwd

    def get_dgrad(self, w, cout_pad=None):
        k = self._key(w)
        if self.key_d != k:
            self.dgrad, self.key_d = pack_weight_dgrad(w, cout_pad), k
        return self.dgrad

    def __deepcopy__(self, memo):
        return _PackCache()


class ModelPacker:
    """Packed "f16x2" weights of ALL plain convolutions of a model, refreshed with two launches (csrc/conv_h2.hip
    pack_h2_multi_kernel) when parameters have changed -- every step for the student (SGD) and the teacher (EMA).  The
    limb planes are persistent buffers rewritten in place (stream order protects their readers); every refresh takes
    fresh zeroed max|w| slots.  Convolutions with padded layouts (stem, class head) keep their own _PackCache path."""

    def __init__(self, convs):
        self.convs = [c for c in convs if c.weight.shape[0] % 32 == 0 and c.weight.shape[1] % 32 == 0]
        self.bufs = {}

    def refresh(self, need_dgrad):
        if CONV_MODE != "f16x2" or not self.convs:
            return
        stale = []
        for c in self.convs:
            w, cache = c.weight, c._pack
            key = _PackCache._key(w)
            if cache.key_f != key or (need_dgrad and cache.key_d != key):
                stale.append((c, key))
        if not stale:
            return
        dev = stale[0][0].weight.device
        ents, biggest = [], 0  # biggest: running total of blocks
        for c, key in stale:
            w = c.weight.detach()
            if not w.is_contiguous():
                w = w.contiguous()
            cout, cin, kh, kw = w.shape
            buf = self.bufs.get(id(c))
            if buf is None or buf[0].device != w.device:
                buf = self.bufs[id(c)] = (torch.empty(2, cout, kh * kw * cin, device=w.device, dtype=torch.float16),
                                          torch.empty(2, cin, kh * kw * cout, device=w.device, dtype=torch.float16))
            slot = amax_slot(w.device, long_lived=True)
            tag_amax(c.weight, slot)
            cache = c._pack
            cache.fwd, cache.key_f = H2Weight(buf[0], slot), key
            if need_dgrad:
                cache.dgrad, cache.key_d = H2Weight(buf[1], slot), key
            else:
                cache.key_d = None
            ents.append(_lib.OndaPackEntry(w.data_ptr(), buf[0].data_ptr(), buf[1].data_ptr() if need_dgrad else None, slot.data_ptr(),
                                           cout, cin, kh * kw, biggest))
            biggest += query("onda_pack_blocks", cout, cin, kh * kw)  # running total: the next entry's first block
            self._keep = getattr(self, "_keep", [])
            self._keep.append(w)
        table = _table(ents, _lib.OndaPackEntry, dev)
        call("onda_pack_weights_h2_multi", _p(table), len(ents), biggest, _stream())
        self._keep = []


class Conv2dFn(torch.autograd.Function):
    """NHWC convolution (+bias) with optional BatchNorm statistic partials from the epilogue."""

    @staticmethod
    def forward(ctx, x, weight, bias, cache, stride, dil, pad, want_stats, cout_pad):
        cout, cin, k, _ = weight.shape
        co = cout_pad or cout
        wp = cache.get_fwd(weight, cout_pad)
        y, stats, _tiles = conv_forward(x, wp, k, stride, dil, pad, co, shift=_pad_vec(bias, co), want_stats=want_stats)
        ctx.save_for_backward(x, weight)
        ctx.xscale = known_amax(x)  # "f16x2": max|x| of the input, reused by the weight gradient
        hit = getattr(x, "_onda_limbs", None)
        ctx.xlimbs = hit[1] if hit is not None and hit[0] == x._version else None  # ... and its limb planes
        ctx.weight_param = weight  # the Parameter itself: its .grad is the accumulation target
        ctx.sink = _sink_of(x) if ctx.needs_input_grad[0] else None
        ctx.cache, ctx.geom, ctx.has_bias = cache, (k, stride, dil, pad, cout, cin, cout_pad), bias is not None
        # the statistics output never has a gradient: do not let autograd build a zero tensor of its shape for backward
        ctx.set_materialize_grads(False)
        if want_stats:
            ctx.mark_non_differentiable(stats)
            return y, stats
        return y, None

    @staticmethod
    def backward(ctx, dy, _dstats):
        if dy is None:
            return (None,) * 9
        x, weight = ctx.saved_tensors
        k, stride, dil, pad, cout, cin, cout_pad = ctx.geom
        if not is_limb_only(dy):
            dy = as_nhwc(dy)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            sink = ctx.sink
            wpd = ctx.cache.get_dgrad(weight, cout_pad)
            if sink is not None and sink.buf is not None:
                conv_dgrad(dy, wpd, k, stride, dil, pad, cin, x.shape[1:3], accumulate=sink.buf)
                _sink_done(sink)
            else:
                dx = _sink_give(sink, conv_dgrad(dy, wpd, k, stride, dil, pad, cin, x.shape[1:3]), True)
        if ctx.needs_input_grad[1]:
            into = _accumulate_target(ctx.weight_param)
            dw = conv_wgrad(x, dy, k, stride, dil, pad, cout, cin, into=into, xscale=ctx.xscale, xlimbs=ctx.xlimbs)
            if into is not None and GRAD_READY is not None:
                GRAD_READY(ctx.weight_param)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = colsum(dy)[:cout]
        return dx, dw, db, None, None, None, None, None, None


def _pad_vec(v, n):
    if v is None or v.numel() == n:
        return v
    out = torch.zeros(n, device=v.device, dtype=v.dtype)
    out[: v.numel()] = v.detach()
    return out


def stem_patches(x_nchw, Ho, Wo, l2):
    """The stem's patch matrix [B,Ho,Wo,STEM_K] of an image batch.  In pre-split "f16x2" mode it exists as limb planes
    only, written by the patch kernel itself (max|patches| = max|image|, one small pass over the image).  The matrix is
    remembered on the image tensor: the teacher, the static model and the student all read the same target batch."""
    hit = getattr(x_nchw, "_onda_patches", None)
    key = (x_nchw._version, bool(l2), Ho, Wo)
    if hit is not None and hit[0] == key:
        return hit[1]
    B, _, H, W = x_nchw.shape
    dev = x_nchw.device
    if l2:
        amax = amax_slot(dev)
        call("onda_absmax", _p(x_nchw), 1, x_nchw.numel(), x_nchw.numel(), _p(amax), _stream())
        M = B * Ho * Wo
        planes = torch.empty(2, M, STEM_K, device=dev, dtype=torch.float16)
        call("onda_stem_im2col_l2", _p(x_nchw), _p(amax), _p(planes), M * STEM_K, B, H, W, Ho, Wo, STEM_K, _stream())
        col = limb_only((B, Ho, Wo, STEM_K), dev, Limbs(planes, amax, STEM_K, M * STEM_K))
    else:
        col = torch.empty(B, Ho, Wo, STEM_K, device=dev, dtype=torch.float32)
        call("onda_stem_im2col", _p(x_nchw), _p(col), B, H, W, Ho, Wo, STEM_K, _stream())
    try:
        x_nchw._onda_patches = (key, col)
    except AttributeError:
        pass
    return col


def stem_prefetch(x_nchw):
    """Build (and cache on the tensor) the stem's patch matrix of an image batch now, on the current stream: several passes
    that read the same batch from different streams then all find it."""
    B, _, H, W = x_nchw.shape
    l2 = CONV_MODE == "f16x2" and H2_PATH == "dma" and STEM_K % 32 == 0
    return stem_patches(x_nchw, conv_out_size(H, 7, 2, 1, 3), conv_out_size(W, 7, 2, 1, 3), l2)


class StemConvFn(torch.autograd.Function):
    """7x7 / stride 2 / pad 3 stem on the NCHW image: im2col patches + the same MFMA GEMM."""

    @staticmethod
    def forward(ctx, x_nchw, weight, cache, want_stats):
        _require_cuda(x_nchw, "image")
        x_nchw = x_nchw.contiguous()
        B, _, H, W = x_nchw.shape
        Ho, Wo = conv_out_size(H, 7, 2, 1, 3), conv_out_size(W, 7, 2, 1, 3)
        wp = cache.get_fwd(weight, None, STEM_K)
        col = stem_patches(x_nchw, Ho, Wo, _use_l2(wp, STEM_K))
        y, stats, _ = conv_forward(col, wp, 1, 1, 1, 0, weight.shape[0], want_stats=want_stats)
        ctx.save_for_backward(col)  # dropped again by autograd when no graph is being recorded
        ctx.col_limbs = limbs_of(col) if is_limb_only(col) else None
        ctx.cout, ctx.weight = weight.shape[0], weight
        ctx.set_materialize_grads(False)
        if want_stats:
            ctx.mark_non_differentiable(stats)
            return y, stats
        return y, None

    @staticmethod
    def backward(ctx, dy, _dstats):
        if dy is None:
            return None, None, None, None
        (col,) = ctx.saved_tensors
        into = _accumulate_target(ctx.weight)
        dw = conv_wgrad(col, as_nhwc(dy), 1, 1, 1, 0, ctx.cout, 3, flat_k=49, into=into, xlimbs=ctx.col_limbs)
        if into is not None and GRAD_READY is not None:
            GRAD_READY(ctx.weight)
        return None, dw, None, None


def stem_eval(x_nchw, weight, cache, scale, shift):
    """Eval-mode stem: patches -> GEMM with folded BatchNorm + ReLU in the epilogue (no graph)."""
    _require_cuda(x_nchw, "image")
    with torch.no_grad():
        x_nchw = x_nchw.contiguous()
        B, _, H, W = x_nchw.shape
        Ho, Wo = conv_out_size(H, 7, 2, 1, 3), conv_out_size(W, 7, 2, 1, 3)
        wp = cache.get_fwd(weight, None, STEM_K)
        col = stem_patches(x_nchw, Ho, Wo, _use_l2(wp, STEM_K))
        y, _, _ = conv_forward(col, wp, 1, 1, 1, 0, weight.shape[0], scale=scale, shift=shift, relu=True)
    return y


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
        amax = amax_slot(y.device) if CONV_MODE == "f16x2" else None  # the output feeds a conv: leave max|out| behind
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
        amax = amax_slot(y.device) if CONV_MODE == "f16x2" else None
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




# BatchNorm's small statistics passes inside the big launches that need them (csrc/norm_l2.hip: bn_finalize_l2 in the apply
# launch, bn_bwd_sums_l2 in the backward apply launch): 156 launches per adaptation step fewer -- and 27 ms per step SLOWER
# (A/B on one box, twice: 94.1 / 94.1 ms with the separate launches, 121.5 / 121.7 ms fused; gpurun_out/r06_a_*):
# the ~2 000 resident workgroups that wait for the first 64 poll ONE line at device scope, across the 8 XCDs' fabric, and the
# finalizing workgroups' own table reads and atomics queue behind the polls (175 us per fused launch against 9 us for the
# launch it removes).  Built, correct (the whole GPU suite passes with it on), OFF by default; ONDA_FUSE_BN=1 /
# tools/ab_flag.py onda_amd.ops FUSE_BN_FINALIZE True -- ... to measure it again.
FUSE_BN_FINALIZE = os.environ.get("ONDA_FUSE_BN", "0") == "1"


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
        if FUSE_BN_FINALIZE and nhwc_ld(y) == C:
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
             int(FUSE_BN_FINALIZE), _stream())
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
        amax = amax_slot(cat.device) if CONV_MODE == "f16x2" else None
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
_TABLE_STAGES = {}


def _table(entries, struct, device):
    """The entry table of a multi-tensor launch on `device`.  A pageable host->device copy would stop the host until
    the stream has drained (measured: 39 ms behind the backward pass for the optimizer's table, 13 ms per weight
    repack): the bytes go through a small ring of pinned staging buffers with an asynchronous copy instead; a buffer is
    reused once the event recorded behind its copy has completed."""
    arr = (struct * len(entries))(*entries)
    raw = bytearray(bytes(arr))
    n = len(raw)
    host = torch.frombuffer(raw, dtype=torch.uint8)
    if torch.device(device).type != "cuda":
        return host.to(device)
    ring = _TABLE_STAGES.setdefault(str(device), [])
    slot = None
    for cand in ring:
        if cand[0].numel() >= n and cand[1].query():
            slot = cand
            break
    if slot is None:
        if len(ring) >= 16:  # (cannot happen with a few tables per step; bound the ring anyway)
            slot = max(ring, key=lambda c: c[0].numel())
            slot[1].synchronize()
            if slot[0].numel() < n:
                slot[0] = torch.empty(n, dtype=torch.uint8, pin_memory=True)
        else:
            slot = [torch.empty(max(n, 1 << 14), dtype=torch.uint8, pin_memory=True), torch.cuda.Event()]
            ring.append(slot)
    slot[0][:n].copy_(host)
    dev = torch.empty(n, dtype=torch.uint8, device=device)
    dev.copy_(slot[0][:n], non_blocking=True)
    slot[1].record()
    return dev


def sgd_multi(items, momentum, weight_decay, grad_scale=1.0):
    """items: list of (param, grad, buf, lr, times, fresh).  One launch for all tensors; every gradient element is
    multiplied by `grad_scale` on the way in."""
    blk, ents, first = query("onda_multi_tensor_block"), [], 0
    for p, g, b, lr, times, fresh in items:
        ents.append(_lib.OndaSgdEntry(p.data_ptr(), g.data_ptr(), b.data_ptr(), p.numel(), float(lr), int(times), int(fresh), first))
        first += -(-p.numel() // blk)
    dev = items[0][0].device
    table = _table(ents, _lib.OndaSgdEntry, dev)
    call("onda_sgd_multi", _p(table), len(ents), float(momentum), float(weight_decay), float(grad_scale), first, _stream())
    for p, *_ in items:
        torch.autograd.graph.increment_version(p)


def ema_multi(items, cache=None):
    """items: list of (k, q, keep, blend): k = k*keep + q*blend, one launch.  `cache` (a dict the caller keeps next to a
    FIXED list of tensors): the device table is rebuilt only when an address or a factor changed -- at the end of a step
    nothing is queued behind the optimizer launch, so the ~1.5 ms of host work that 320 table entries cost were 1.5 ms of
    idle device per step (tools/trace_gaps.py)."""
    key = [(k.data_ptr(), q.data_ptr(), keep, blend) for k, q, keep, blend in items]
    if cache is not None and cache.get("key") == key:
        table, n, first = cache["table"], cache["n"], cache["blocks"]
    else:
        blk, ents, first = query("onda_multi_tensor_block"), [], 0
        for k, q, keep, blend in items:
            ents.append(_lib.OndaEmaEntry(k.data_ptr(), q.data_ptr(), k.numel(), float(keep), float(blend), first))
            first += -(-k.numel() // blk)
        table, n = _table(ents, _lib.OndaEmaEntry, items[0][0].device), len(ents)
        if cache is not None:
            cache.update(key=key, table=table, n=n, blocks=first)
    call("onda_ema_multi", _p(table), n, first, _stream())
    torch.autograd.graph.increment_version([k for k, *_ in items])
