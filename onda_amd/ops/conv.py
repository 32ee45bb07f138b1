"""Convolutions: weight packers, gradient sinks, forward / data gradient / weight gradient launches (f16x2 pre-split and exact
fp32), the per-model packer, the autograd functions of the convs and the stem."""
import ctypes
import os
from ctypes import byref

import torch

from .. import _lib
from .._lib import OndaConv, OndaLimbOut, call, query
from . import _state
from ._state import BN_EPS, GN_EPS, GN_GROUPS, HEAD_PAD, STEM_K
from .core import _conv_ws, _desc, _group_split, _l2_name, _launch, _p, _require_cuda, _stream, as_nhwc, conv_out_size, nhwc_ld
from .limbs import H2Weight, Limbs, amax_slot, is_limb_only, known_amax, limb_only, limbs_of, tag_amax
from .tables import _table


def _use_l2(wp, cin):
    return isinstance(wp, H2Weight) and _state.H2_PATH == "dma" and cin % 32 == 0


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
    if _state.CONV_MODE == "f16x2" and (kp % 32 == 0 if flat else cin % 32 == 0):
        return _pack_h2(weight, cout_pad, kp, 0, cout_pad)
    dst = torch.empty(cout_pad, kp, device=weight.device, dtype=torch.float32)
    call("onda_pack_weight_fwd", _p(weight.detach().contiguous()), _p(dst), cout, cin, taps, cout_pad, kp, _stream())
    return dst


def pack_weight_dgrad(weight, cout_pad=None):
    """OIHW -> [Cin][taps (flipped)][Cout_pad]: the data gradient of a stride-1 conv is a conv of dy with this."""
    cout, cin, kh, kw = weight.shape
    cout_pad = cout_pad or cout
    if _state.CONV_MODE == "f16x2" and cout_pad % 32 == 0:  # (the data gradient's K runs over the output channels)
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




def share_grad(x):
    """Mark `x` (about to be consumed by several of this module's Functions, and by nothing else) for GradSink."""
    if _state.SHARE_GRADS and torch.is_grad_enabled() and x.requires_grad and getattr(x, "_onda_sink", None) is None:
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
                 else query("onda_conv_tiles_mc", B * Ho * Wo, cout))
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
    l2 = _state.CONV_MODE == "f16x2" and Cin % 32 == 0 and Co % 32 == 0
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
        return (w.data_ptr(), w._version, w.device, _state.CONV_MODE)

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
        if _state.CONV_MODE != "f16x2" or not self.convs:
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
            if into is not None and _state.GRAD_READY is not None:
                _state.GRAD_READY(ctx.weight_param)
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
    l2 = _state.CONV_MODE == "f16x2" and _state.H2_PATH == "dma" and STEM_K % 32 == 0
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
        if into is not None and _state.GRAD_READY is not None:
            _state.GRAD_READY(ctx.weight)
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


