"""Input pipeline on the GPU: what `Segmentation_db.__getitem__` does to a DECODED sample
(reference `framework/dataset/segmentation_db.py:56-99`, `_load_img` `base_dataset.py:89-95`,
`color_mapper` `utils/func.py:88-115`), as three HIP kernels (csrc/pipeline.hip).

PNG decoding stays where the reference has it (PIL, loader processes); everything after it --
the antialiased BICUBIC resize, BGR flip, ToTensor + Normalize, the two NEAREST label resizes and
the id map -- runs on the device on the raw uint8 frames, so a loader worker ships 6 MB + 2 MB of
bytes per 2048x1024 sample instead of resizing on the CPU.  Results are bit-identical to the
reference's path (fixture G9, tests/test_hip_pipeline.py).

The coefficient tables are Pillow's (src/libImaging/Resample.c precompute_coeffs +
normalize_coeffs_8bpc, Geometry.c ImagingScaleAffine), evaluated in float64 in the same operation
order, once per (input size, output size).
"""
import ctypes
import math

import numpy as np
import torch

from ._lib import call
from .ops import _p, _stream

PRECISION_BITS = 32 - 8 - 2


def bicubic_tables(in_size, out_size):
    """(bounds int32[out, 2], kk int32[out, ksize], ksize) of Pillow's BICUBIC resample of a full axis."""
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    center = (np.arange(out_size, dtype=np.float64) + 0.5) * scale
    xmin = np.maximum(np.trunc(center - support + 0.5), 0).astype(np.int64)
    xmax = np.minimum(np.trunc(center + support + 0.5), in_size).astype(np.int64)
    n = xmax - xmin
    ss = 1.0 / filterscale
    j = np.arange(ksize, dtype=np.int64)[None, :]
    x = np.abs((j + xmin[:, None] - center[:, None] + 0.5) * ss)
    a = -0.5
    w = np.where(x < 1.0, ((a + 2.0) * x - (a + 3.0)) * x * x + 1, np.where(x < 2.0, (((x - 5) * x + 8) * x - 4) * a, 0.0))
    w = np.where(j < n[:, None], w, 0.0)
    ww = np.zeros(out_size, np.float64)
    for c in range(ksize):  # the same left-to-right accumulation as the C loop
        ww = ww + w[:, c]
    w = np.where(ww[:, None] != 0.0, w / np.where(ww == 0.0, 1.0, ww)[:, None], w)
    q = w * float(1 << PRECISION_BITS)
    kk = np.where(w < 0, np.trunc(-0.5 + q), np.trunc(0.5 + q)).astype(np.int32)
    kk = np.where(j < n[:, None], kk, 0).astype(np.int32)
    bounds = np.stack([xmin, n], axis=1).astype(np.int32)
    return bounds, kk, ksize


def nearest_table(in_size, out_size):
    """Source index per output coordinate of `Image.resize(..., NEAREST)`: (int) of a running double sum."""
    a = float(in_size) / out_size
    steps = np.full(out_size, a, np.float64)
    steps[0] = a * 0.5
    xo = np.cumsum(steps)  # sequential accumulation, like `xo += a`
    return np.clip(np.where(xo < 0.0, -1, np.trunc(xo)), 0, in_size - 1).astype(np.int32)


class GpuPreprocessor:
    """`Segmentation_db(...)`'s per-sample transform for decoded frames already on the device.

    image_size / labels_size: (W, H) as in the reference's constructor; mean / std in [0, 255] and in
    the channel order they are applied in (to the BGR-flipped image, `base_transform(mean, std)`);
    `id_map`: 256-entry table source id -> train id (`color_mapper` of an id-valued map).
    """

    def __init__(self, image_size, labels_size=None, mean=(0.0, 0.0, 0.0), std=(255.0, 255.0, 255.0), id_map=None,
                 device="cuda:0"):
        self.image_size = tuple(int(v) for v in image_size)
        self.labels_size = tuple(int(v) for v in (labels_size or image_size))
        self.device = torch.device(device)
        self.mean = (np.asarray(mean, np.float64) / 255).astype(np.float32)  # Normalize(mean / 255, std / 255)
        self.std = (np.asarray(std, np.float64) / 255).astype(np.float32)
        lut = np.arange(256) if id_map is None else np.asarray(id_map)
        full = np.zeros(256, np.uint8)
        full[: min(256, len(lut))] = np.asarray(lut[:256]).astype(np.uint8)
        self.lut = torch.from_numpy(full).to(self.device)
        self._tables = {}

    def _bicubic(self, n_in, n_out):
        key = ("b", n_in, n_out)
        if key not in self._tables:
            b, k, ks = bicubic_tables(n_in, n_out)
            self._tables[key] = (torch.from_numpy(b).to(self.device), torch.from_numpy(k).to(self.device), ks)
        return self._tables[key]

    def _nearest(self, n_in, n_out):
        key = ("n", n_in, n_out)
        if key not in self._tables:
            self._tables[key] = torch.from_numpy(nearest_table(n_in, n_out)).to(self.device)
        return self._tables[key]

    def image(self, rgb_u8):
        """u8[H0, W0, 3] RGB frame on the device -> f32[3, H, W] (BGR, normalised): `output["image"]`."""
        if rgb_u8.dtype != torch.uint8 or rgb_u8.dim() != 3 or rgb_u8.shape[2] != 3 or not rgb_u8.is_cuda:
            raise RuntimeError("onda_amd.pipeline: image must be a uint8 [H, W, 3] tensor on the GPU")
        rgb_u8 = rgb_u8.contiguous()
        H0, W0, _ = rgb_u8.shape
        W, H = self.image_size
        bh, kh, ksh = self._bicubic(W0, W)
        bv, kv, ksv = self._bicubic(H0, H)
        tmp = torch.empty(H0, W, 3, device=rgb_u8.device, dtype=torch.uint8)
        call("onda_resample_h_u8", _p(rgb_u8), _p(tmp), H0, W0, W, _p(bh), _p(kh), ksh, _stream())
        out = torch.empty(3, H, W, device=rgb_u8.device, dtype=torch.float32)
        m = (ctypes.c_float * 3)(*self.mean.tolist())
        s = (ctypes.c_float * 3)(*self.std.tolist())
        call("onda_resample_v_norm", _p(tmp), _p(out), H0, W, H, _p(bv), _p(kv), ksv, m, s, 1, _stream())
        return out

    def _label(self, lab_u8, size):
        H0, W0 = lab_u8.shape
        W, H = size
        out = torch.empty(H, W, device=lab_u8.device, dtype=torch.uint8)
        call("onda_resize_nearest_lut", _p(lab_u8), _p(out), W0, H, W, _p(self._nearest(W0, W)), _p(self._nearest(H0, H)),
             _p(self.lut), _stream())
        return out

    def labels(self, lab_u8):
        """u8[H0, W0] id image on the device -> (`label` u8[H, W], `label_res` u8[H/8+1, W/8+1])."""
        if lab_u8.dtype != torch.uint8 or lab_u8.dim() != 2 or not lab_u8.is_cuda:
            raise RuntimeError("onda_amd.pipeline: label must be a uint8 [H, W] tensor on the GPU")
        lab_u8 = lab_u8.contiguous()
        W, H = self.labels_size
        return self._label(lab_u8, (W, H)), self._label(lab_u8, (int(W / 8 + 1), int(H / 8 + 1)))

    def batch(self, frames, label_frames=None):
        """Lists of decoded frames -> the batch dict fields the adaptation step reads
        (`image` f32[B,3,H,W], `label` u8[B,H,W], `label_res` u8[B,H/8+1,W/8+1])."""
        out = {"image": torch.stack([self.image(f.to(self.device, non_blocking=True)) for f in frames])}
        if label_frames is not None:
            pairs = [self.labels(f.to(self.device, non_blocking=True)) for f in label_frames]
            out["label"] = torch.stack([p[0] for p in pairs])
            out["label_res"] = torch.stack([p[1] for p in pairs])
        return out
