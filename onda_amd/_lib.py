"""ctypes binding of libonda_hip.so (the C ABI declared in include/onda_hip.h).

The library is the product: if it is missing or a symbol is absent this module raises --
there is no eager / CPU fallback anywhere in ``onda_amd``.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int64, c_void_p

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ONDA_LIB_PATH") or os.path.join(HERE, "libonda_hip.so")  # override: diagnostic builds

P = c_void_p  # device pointers travel as integers (tensor.data_ptr())
I, L, F = c_int, c_int64, c_float


class OndaConv(Structure):
    _fields_ = [(n, c_int) for n in ("B Hi Wi Cin Ho Wo Cout kh kw stride dil pad ldx ldy ldr out_os Hf Wf relu").split()] + \
               [("run_if", c_void_p), ("stat_split", c_int64), ("plain_schedule", c_int), ("pix_table", c_void_p),
                ("pix_stride", c_int64)]


class OndaSwitchCfg(Structure):
    _fields_ = [("limit", c_int), ("level_kind", c_int), ("use_exp", c_int), ("pad_", c_int), ("exp_const", ctypes.c_double),
                ("one_minus_exp_const", ctypes.c_double), ("taps_total", ctypes.c_double), ("gray_lo", ctypes.c_double),
                ("gray_hi", ctypes.c_double), ("dev_threshold", ctypes.c_double)]


class OndaLimbOut(Structure):
    _fields_ = [("out", c_void_p), ("out_plane", c_int64), ("out_bound", c_void_p), ("out_amax", c_void_p), ("kb", c_void_p),
                ("xtrue", c_void_p), ("res", c_void_p), ("res_plane", c_int64), ("res_amax", c_void_p), ("res_true", c_void_p)]


class OndaSgdEntry(Structure):
    _fields_ = [("p", c_void_p), ("g", c_void_p), ("buf", c_void_p), ("n", c_int64), ("lr", c_float),
                ("times", c_int), ("fresh", c_int), ("first_block", c_int)]


class OndaPackEntry(Structure):
    _fields_ = [("w", c_void_p), ("fwd", c_void_p), ("dgrad", c_void_p), ("amax", c_void_p), ("Cout", c_int), ("Cin", c_int),
                ("taps", c_int), ("first_block", c_int)]


class OndaEmaEntry(Structure):
    _fields_ = [("k", c_void_p), ("q", c_void_p), ("n", c_int64), ("keep", c_float), ("blend", c_float), ("first_block", c_int)]


# name -> (restype, argtypes); mirrors include/onda_hip.h one to one
SIGNATURES = {
    "onda_conv_tiles_m": (I, [I]),
    "onda_conv_tiles_mc": (I, [I, I]),
    "onda_conv_ws_floats": (L, []),
    "onda_conv2d_fwd": (I, [P, P, P, P, P, P, P, P, POINTER(OndaConv), P]),
    "onda_absmax": (I, [P, L, I, I, P, P]),
    "onda_pack_weight_h2": (I, [P, P, I, I, I, I, I, I, I, P, P]),
    "onda_pack_weights_h2_multi": (I, [P, I, L, P]),
    "onda_pack_blocks": (I, [I, I, I]),
    "onda_split_h2": (I, [P, L, I, I, P, I, L, P, P]),
    "onda_stem_im2col_l2": (I, [P, P, P, L, I, I, I, I, I, I, P]),
    "onda_debug_stamps": (None, [P]),
    "onda_conv2d_fwd_l2_limbs": (I, [P, L, P, P, P, P, P, P, P, P, P]),
    "onda_switch_state_doubles": (I, [I]),
    "onda_switch_max_window": (I, []),
    "onda_switch_step": (I, [P, P, P, I, P, POINTER(OndaSwitchCfg), P, P]),
    "onda_select_prior": (I, [P, P, F, P, F, P, L, P]),
    "onda_gate_scalar": (I, [P, P, P, P]),
    "onda_conv_l2_variant": (I, [L, I]),
    "onda_conv_l2_kernel_id": (I, [L, I, I, I]),
    "onda_conv_l2_tiles_m": (I, [L, I, I, I]),
    "onda_conv_l2_tiles_m_split": (I, [L, I, I, I, L, I, POINTER(c_int)]),
    "onda_conv_l2_live_fraction": (ctypes.c_double, [POINTER(OndaConv), I]),
    "onda_conv_wgrad_l2_live_fraction": (ctypes.c_double, [POINTER(OndaConv), I]),
    "onda_conv_wgrad_l2_variant": (I, [I, I]),
    "onda_conv2d_wgrad_l2_table_stride": (L, [POINTER(OndaConv)]),
    "onda_conv2d_wgrad_l2_table": (I, [POINTER(OndaConv), P, P]),
    "onda_conv2d_wgrad_l2": (I, [P, L, P, P, L, P, P, I, I, POINTER(OndaConv), P]),
    "onda_conv2d_fwd_l2": (I, [P, L, P, P, P, P, P, P, P, P, I, P, P, POINTER(OndaConv), P]),
    "onda_bn_finalize_l2": (I, [P, I, I, L, F, P, P, P, P, P, F, P, P, P, I, P, P, L, I, I, P, I, P]),
    "onda_bn_apply_l2": (I, [P, P, P, P, P, P, L, P, P, L, P, L, I, I, P, L, P]),
    "onda_bn_bwd_l2_ws": (L, [L, I]),
    "onda_bn_train_l2": (I, [P, P, I, F, P, P, P, P, P, F, P, P, P, P, I, P, P, P, L, I, P, L, I, I, P]),
    "onda_bn_bwd_l2": (I, [P, P, L, P, P, P, P, P, P, L, P, P, P, L, I, I, P, L, I, P]),
    "onda_conv2d_wgrad": (I, [P, P, P, I, I, POINTER(OndaConv), P]),
    "onda_wgrad_reduce": (I, [P, P, I, I, I, I, I, I, I, I, P]),
    "onda_pack_weight_fwd": (I, [P, P, I, I, I, I, I, P]),
    "onda_pack_weight_dgrad": (I, [P, P, I, I, I, I, P]),
    "onda_stem_im2col": (I, [P, P, I, I, I, I, I, I, P]),
    "onda_bn_finalize": (I, [P, I, I, L, F, P, P, P, P, P, F, P]),
    "onda_bn_stats": (I, [P, L, I, I, P, POINTER(c_int), P]),
    "onda_bn_apply": (I, [P, P, P, P, P, P, P, L, I, I, P, P]),
    "onda_bn_fold": (I, [P, P, P, P, F, P, P, I, P]),
    "onda_bn_bwd_ws": (L, [L, I]),
    "onda_bn_bwd": (I, [P, P, P, P, P, P, P, P, P, L, I, I, P, P]),
    "onda_gn_ws": (L, [I, L, I]),
    "onda_gn_fwd": (I, [P, I, P, P, P, P, I, P, P, P, I, L, I, I, F, I, P, P]),
    "onda_gn_bwd": (I, [P, I, P, I, P, I, P, P, P, P, P, P, P, P, I, L, I, I, I, P]),
    "onda_maxpool_fwd": (I, [P, P, P, I, I, I, I, I, I, P]),
    "onda_maxpool_fwd_limbs": (I, [P, P, P, P, I, I, I, I, I, I, P]),
    "onda_maxpool_bwd": (I, [P, P, P, I, I, I, I, I, I, P]),
    "onda_colsum_ws": (L, [I, L, I]),
    "onda_colsum": (I, [P, I, P, I, P, F, P, I, L, I, P]),
    "onda_se_fc_fwd": (I, [P, P, P, P, P, P, P, I, I, I, P]),
    "onda_se_fc_bwd": (I, [P, P, P, P, P, P, P, P, P, P, P, P, F, I, I, I, P]),
    "onda_chan_scale": (I, [P, P, P, P, I, L, I, P]),
    "onda_chan_scale_limbs": (I, [P, P, P, L, P, I, L, I, P]),
    "onda_upsample_fwd": (I, [P, I, P, I, I, I, I, I, I, P]),
    "onda_upsample_bwd": (I, [P, P, I, I, I, I, I, I, I, P]),
    "onda_upsample_argmax": (I, [P, I, P, I, I, I, I, I, I, P]),
    "onda_upsample_ce_ws": (L, [I, I, I]),
    "onda_upsample_ce_fwd": (I, [P, I, P, P, P, I, I, I, I, I, I, P]),
    "onda_upsample_ce_bwd_ws": (L, [I, I, I, I]),
    "onda_upsample_ce_bwd": (I, [P, I, P, P, P, F, P, P, I, I, I, I, I, I, P]),
    "onda_upsample_argmax_hist": (I, [P, I, P, P, P, I, I, I, I, I, I, P]),
    "onda_softmax_stats": (I, [P, I, P, I, P, P, P, L, I, P]),
    "onda_seg_loss_fwd": (I, [P, I, P, P, P, L, I, P]),
    "onda_seg_loss_bwd": (I, [P, I, P, P, P, F, F, F, P, L, I, P]),
    "onda_proto_sigma": (I, [P, P, P, P, I, I, P]),
    "onda_proto_assign_blocks": (I, [L]),
    "onda_proto_assign": (I, [P, I, P, I, P, P, I, F, F, P, P, P, P, L, I, I, P]),
    "onda_proto_distances": (I, [P, I, P, P, I, P, L, I, I, P]),
    "onda_proto_sums_ws": (L, [L, I, I]),
    "onda_proto_class_sums": (I, [P, I, P, P, P, P, L, I, I, P]),
    "onda_proto_ema": (I, [P, P, P, P, F, I, I, P]),
    "onda_proto_append": (I, [P, P, P, P, P, I, I, P]),
    "onda_sgd_multi": (I, [P, I, F, F, F, L, P]),
    "onda_ema_multi": (I, [P, I, L, P]),
    "onda_multi_tensor_block": (I, []),
    "onda_resample_h_u8": (I, [P, P, I, I, I, P, P, I, P]),
    "onda_resample_v_norm": (I, [P, P, I, I, I, P, P, I, POINTER(c_float), POINTER(c_float), I, P]),
    "onda_resize_nearest_lut": (I, [P, P, I, I, I, P, P, P, P]),
    "onda_version": (c_char_p, []),
    "onda_limb2_scale": (F, []),
}

_lib = None


class OndaLibraryError(RuntimeError):
    pass


def load():
    """dlopen the in-tree library once and attach the prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OndaLibraryError(
            f"{LIB_PATH} is missing: build it with `python -m onda_amd.build` (hipcc --offload-arch=gfx950). "
            "onda_amd has no fallback path.")
    # PyTorch-ROCm ships its own libamdhip64.so.7; importing torch first makes that copy THE HIP
    # runtime of the process, so our kernels and torch's tensors/streams share one device context
    # (loaded the other way round, the two runtimes disagree and launches fail with hipErrorNoDevice)
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise OndaLibraryError(f"{LIB_PATH} does not export {name}") from e
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def call(name, *args):
    """Invoke an int-returning entry point; non-zero -> RuntimeError."""
    rc = getattr(load(), name)(*args)
    if rc != 0:
        kind = {-1: "ONDA_EINVAL (unsupported shape / null pointer)", -2: "ONDA_EALIGN (16-byte alignment)"}.get(
            rc, f"hipError_t {rc}")
        raise RuntimeError(f"{name} failed: {kind}")


def query(name, *args):
    """Invoke a value-returning helper (workspace sizes, tile counts)."""
    return getattr(load(), name)(*args)
