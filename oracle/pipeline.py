"""CPU restatement of the reference's input pipeline (TEST INFRASTRUCTURE: only tests/, smoke() and
bench.py's cpu_baseline leg may import this package).

Path restated -- `framework/dataset/segmentation_db.py:56-99` with `framework/dataset/base_dataset.py:89-95`
(`_load_img`) and `framework/utils/func.py:88-115` (`color_mapper`):

    image      = Image.open(f).convert("RGB").resize((W, H), Image.BICUBIC)      -> u8[H, W, 3]
    image      = image[:, :, ::-1]                                                 (RGB -> BGR)
    image      = ToTensor()(image); Normalize(mean / 255, std / 255)(image)       -> f32[3, H, W]
    label      = map(Image.open(f).resize((W, H), Image.NEAREST))                 -> u8[H, W]
    label_res  = map(Image.open(f).resize((W/8+1, H/8+1), Image.NEAREST))         -> u8[H/8+1, W/8+1]

The arithmetic lives in two third-party dependencies that are not under /root/reference:

  * Pillow (pinned `pillow=9.0.1`, environment.yml:89; 12.2.0 in this image -- the resampling code
    is unchanged between them).  Restated here from its published algorithm, `src/libImaging/Resample.c`
    (`precompute_coeffs`, `normalize_coeffs_8bpc`, `ImagingResampleHorizontal_8bpc`,
    `ImagingResampleVertical_8bpc`: separable, antialiased -- the filter support grows with the
    downscale factor --, 22-bit fixed-point coefficients, a uint8 intermediate image after the
    horizontal pass) and `src/libImaging/Geometry.c` (`ImagingScaleAffine`, the NEAREST path of
    `Image.resize`: source index = (int)(running double sum of the scale)).  PINNED bit-exactly by
    fixture G9, which tests/golden/make_golden.py captured by calling the reference's own `_load_img`
    (i.e. Pillow) on PNG files.
  * torchvision (pinned 0.8.2, environment.yml:119; absent here): `ToTensor` = `u8 -> f32, / 255`,
    `Normalize` = `(x - f32(mean)) / f32(std)`, both fp32, restated from its published source
    (`transforms/functional.py: to_tensor, normalize`).  That part of G9 is the formula itself
    evaluated with torch ops -- parity for it is UNPINNED (no torchvision to run).
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2  # Resample.c


def _bicubic(x):
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def resample_coeffs(in_size, out_size, support=2.0, filt=_bicubic):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for the full box (in0 = 0, in1 = in_size).
    Returns (bounds int32[out, 2] = (first source index, count), kk int32[out, ksize])."""
    scale = filterscale = float(in_size) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = support * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = [filt((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        if ww != 0.0:
            w = [v / ww for v in w]
        for x, v in enumerate(w):
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _clip8(acc):
    return np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)


def _resample_axis1(img, bounds, kk):
    """Fixed-point pass along axis 1 of u8[H, W, C] (ImagingResampleHorizontal_8bpc)."""
    H, _, C = img.shape
    out = np.empty((H, bounds.shape[0], C), np.uint8)
    src = img.astype(np.int64)
    for xx in range(bounds.shape[0]):
        x0, n = int(bounds[xx, 0]), int(bounds[xx, 1])
        acc = (src[:, x0:x0 + n, :] * kk[xx, :n].astype(np.int64)[None, :, None]).sum(axis=1) + (1 << (PRECISION_BITS - 1))
        out[:, xx, :] = _clip8(acc)
    return out


def resize_bicubic_u8(img, size):
    """`Image.fromarray(img).resize(size, Image.BICUBIC)` for u8[H, W, C]; size = (W, H).
    Horizontal pass first, then vertical, each only when that axis changes (Resample.c ImagingResampleInner)."""
    W, H = size
    out = img
    if W != img.shape[1]:
        out = _resample_axis1(out, *resample_coeffs(img.shape[1], W))
    if H != img.shape[0]:
        out = _resample_axis1(out.transpose(1, 0, 2), *resample_coeffs(img.shape[0], H)).transpose(1, 0, 2)
    return np.ascontiguousarray(out)


def nearest_table(in_size, out_size):
    """Geometry.c ImagingScaleAffine: source index of each output coordinate; the centre coordinate is
    a RUNNING double sum (xo += a), not a product."""
    a = float(in_size) / out_size
    xo = a * 0.5
    tab = np.empty(out_size, np.int32)
    for x in range(out_size):
        xin = -1 if xo < 0.0 else int(xo)
        tab[x] = min(max(xin, 0), in_size - 1)  # in range for a full-box resize; clamped for safety
        xo += a
    return tab


def resize_nearest(img, size):
    """`Image.fromarray(img).resize(size, Image.NEAREST)`; size = (W, H)."""
    W, H = size
    if (W, H) == (img.shape[1], img.shape[0]):
        return img.copy()
    xt, yt = nearest_table(img.shape[1], W), nearest_table(img.shape[0], H)
    return np.ascontiguousarray(img[yt][:, xt])


def preprocess_image(rgb_u8, size, mean, std):
    """segmentation_db.py:82-83,98-99 with base_transform (:11-13): f32[3, H, W] in BGR order."""
    img = resize_bicubic_u8(rgb_u8, size)[:, :, ::-1]
    t = img.transpose(2, 0, 1).astype(np.float32) / np.float32(255)
    m = (np.asarray(mean, np.float64) / 255).astype(np.float32)[:, None, None]
    s = (np.asarray(std, np.float64) / 255).astype(np.float32)[:, None, None]
    return ((t - m) / s).astype(np.float32)


def labels(label_u8, labels_size, lut):
    """segmentation_db.py:70-76, 85-96 for an id-valued (non-RGB) label image and an id -> train-id table
    (color_mapper with rgb False, func.py:105-115): (label u8[H, W], label_res u8[H/8+1, W/8+1])."""
    lut = np.asarray(lut)
    W, H = labels_size
    full = lut[resize_nearest(label_u8, (W, H)).astype(np.int32)].astype(np.uint8)
    res = lut[resize_nearest(label_u8, (int(W / 8 + 1), int(H / 8 + 1))).astype(np.int32)].astype(np.uint8)
    return full, res
