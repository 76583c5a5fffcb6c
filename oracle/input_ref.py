"""CPU restatement of the reference's training-time image transform -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Reference: train.py:108-116 builds

    Resize([128*S/112]*2) -> RandomCrop([S, S]) -> RandomHorizontalFlip() -> ToTensor() -> Normalize(RGB_MEAN, RGB_STD)

from torchvision and applies it per sample to a PIL image in dataset.py:68-91.  On PIL images torchvision's Resize is
``img.resize((w, h), Image.BILINEAR)``; that arithmetic lives in Pillow (third-party, not under /root/reference;
Pillow 12.2.0 is installed in this image), ``src/libImaging/Resample.c``:

  * ``precompute_coeffs``: per output index the window [xmin, xmin+n) and the triangle-filter weights (support =
    max(1, in/out)), normalised to sum 1 in double precision;
  * ``normalize_coeffs_8bpc``: weights -> int, ``(int)(+-0.5 + w * 2**22)``;
  * ``ImagingResampleHorizontal_8bpc`` then ``...Vertical_8bpc``: ``clip8((2**21 + sum(pixel * k)) >> 22)``, the
    horizontal pass writing a uint8 image that the vertical pass reads (a pass whose size does not change is skipped).

``resize_tables`` / ``resize_u8`` restate that and are pinned against Pillow itself (tests/test_input_pipeline.py runs the
installed Pillow on random and constant images, up- and down-scaling, odd sizes) -- bit-exact for aspect ratios up
to 16:1 (the range tested; beyond ~100:1 the installed Pillow was observed to run the vertical pass first, which this
restatement does not model and the product refuses).  ToTensor / Normalize are
restated from torchvision's documented behaviour (uint8 -> float32, ``div(255)``, ``sub(mean).div(std)`` in float32);
torchvision is not installed here, so that part is PARITY UNPINNED against torchvision itself (the float32 operations are
torch's own).  Crop offsets and flip decisions are inputs: the random stream of torchvision is not reproduced.
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2  # Resample.c


def _triangle(x):
    x = -x if x < 0.0 else x
    return 1.0 - x if x < 1.0 else 0.0


def resize_tables(in_size, out_size):
    """(bounds int32 [out, 2] = (first input index, tap count), coeffs int32 [out, ksize]) of one axis."""
    scale = filterscale = float(in_size) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    coeffs = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        n = xmax - xmin
        w = [_triangle((x + xmin - center + 0.5) * ss) for x in range(n)]
        ww = 0.0
        for v in w:
            ww += v
        if ww != 0.0:
            w = [v / ww for v in w]
        bounds[xx] = (xmin, n)
        for x, v in enumerate(w):
            coeffs[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
    return bounds, coeffs


def _pass(img, bounds, coeffs, axis):
    """One resampling pass of a uint8 [H, W, C] image along ``axis`` (0 = vertical, 1 = horizontal)."""
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((bounds.shape[0],) + src.shape[1:], np.uint8)
    for o in range(bounds.shape[0]):
        lo, n = int(bounds[o, 0]), int(bounds[o, 1])
        acc = np.full(src.shape[1:], 1 << (PRECISION_BITS - 1), np.int64)
        for t in range(n):
            acc += src[lo + t] * int(coeffs[o, t])
        out[o] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def resize_u8(img, out_h, out_w):
    """PIL ``Image.resize((out_w, out_h), BILINEAR)`` on a uint8 [H, W, C] array: horizontal pass, then vertical."""
    h, w = img.shape[:2]
    if w != out_w:
        img = _pass(img, *resize_tables(w, out_w), axis=1)
    if h != out_h:
        img = _pass(img, *resize_tables(h, out_h), axis=0)
    return img


def normalize_lut(mean, std):
    """float32 [256, 3]: ToTensor (``/255``) then Normalize (``(x - mean) / std``), each step rounded to float32."""
    v = np.arange(256, dtype=np.float32)[:, None] / np.float32(255.0)
    return ((v - np.asarray(mean, np.float32)[None, :]) / np.asarray(std, np.float32)[None, :]).astype(np.float32)


def train_transform(img, size, crop_xy, flip, mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5)):
    """uint8 [H, W, 3] -> float32 [3, size, size]: resize to 128*size/112, crop at ``crop_xy`` = (x0, y0), optional
    horizontal flip, ToTensor, Normalize (train.py:108-116)."""
    big = int(128 * size / 112)
    r = resize_u8(img, big, big)
    x0, y0 = crop_xy
    c = r[y0:y0 + size, x0:x0 + size]
    if flip:
        c = c[:, ::-1]
    lut = normalize_lut(mean, std)
    out = np.empty((3, size, size), np.float32)
    for ch in range(3):
        out[ch] = lut[c[:, :, ch], ch]
    return out
