"""GPU-side training-input pipeline.

The reference transforms every sample on a host worker (dataset.py:85-88 applying the Compose of train.py:108-116):
PIL resize to 128x128, random 112x112 crop, random flip, ToTensor, Normalize -> a float32 [3,112,112] tensor per image
(150 KB) that is collated and copied to the GPU.  At >= 14 k images/s per GPU that host work cannot feed eight GPUs.
Here the workers only DECODE: a sample is the uint8 HWC image as stored (37 KB at 112x112, a quarter of the bytes over
PCIe), the batch is staged as uint8 [B,H,W,3], and ONE HIP launch (``fr_augment_u8``) produces the float32 NCHW batch:
Pillow's 8-bit bilinear resample restated bit-exactly, evaluated only inside each image's crop window, flip, and the
ToTensor/Normalize arithmetic as a 256-entry table per channel.  Crop offsets and flips are drawn on the host (the
reference's random stream -- torchvision's use of the global torch RNG -- is not reproduced; the draws are uniform
over the same ranges).

Scope of the bit-exactness claim: Pillow resamples horizontally, then vertically -- except for extreme aspect ratios
(observed with the installed Pillow 12.2 only beyond ~100:1, e.g. 500x3 -> 128x128, where it runs the vertical pass first
and the uint8 rounding of the intermediate image differs).  Staged images whose aspect ratio exceeds ``MAX_ASPECT`` = 16
are refused instead of being transformed with unverified rounding; aligned face crops are square.

``resize_tables`` is Pillow's ``precompute_coeffs`` + ``normalize_coeffs_8bpc`` for the bilinear filter
(``src/libImaging/Resample.c``); tests pin it against the installed Pillow through ``oracle/input_ref.py``.
"""
import math

import numpy as np
import torch

from . import ops

PRECISION_BITS = 32 - 8 - 2
MAX_ASPECT = 16  # largest staged-image aspect ratio the transform accepts (see the module docstring)


def _axis_table(in_size, out_size):
    """int32 [out_size, ksize + 2]: (first input index, tap count, integer weights) per output index."""
    scale = filterscale = float(in_size) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = filterscale  # bilinear: support 1.0, stretched when shrinking
    ksize = int(math.ceil(support)) * 2 + 1
    tab = np.zeros((out_size, ksize + 2), np.int32)
    inv = 1.0 / filterscale
    one = float(1 << PRECISION_BITS)
    for o in range(out_size):
        center = (o + 0.5) * scale
        lo = max(int(center - support + 0.5), 0)
        hi = min(int(center + support + 0.5), in_size)
        weights = []
        total = 0.0
        for i in range(lo, hi):
            d = abs((i - center + 0.5) * inv)
            w = 1.0 - d if d < 1.0 else 0.0
            weights.append(w)
            total += w
        tab[o, 0], tab[o, 1] = lo, hi - lo
        for j, w in enumerate(weights):
            if total != 0.0:
                w = w / total
            tab[o, 2 + j] = int(w * one - 0.5) if w < 0 else int(w * one + 0.5)
    return tab


def resize_tables(in_h, in_w, out_h, out_w):
    """(xtab [out_w, kx+2], ytab [out_h, ky+2]) for ``Image.resize((out_w, out_h), BILINEAR)`` of an in_h x in_w image.
    An axis whose size does not change gets the identity table (one tap of weight 2**22), which is what skipping that
    pass amounts to."""
    return _axis_table(in_w, out_w), _axis_table(in_h, out_h)


def normalize_lut(mean, std):
    """float32 [256, 3]: uint8 value -> ((v / 255) - mean) / std, rounded to float32 after every step (ToTensor,
    Normalize)."""
    v = np.arange(256, dtype=np.float32)[:, None] / np.float32(255.0)
    return ((v - np.asarray(mean, np.float32)[None, :]) / np.asarray(std, np.float32)[None, :]).astype(np.float32)


class GpuTrainTransform(object):
    """Batch form of the reference's train transform on staged uint8 images.

        tf = GpuTrainTransform(112, RGB_MEAN, RGB_STD)
        x = tf(u8_batch.to(device, non_blocking=True))        # uint8 [B,H,W,3] -> float32 [B,3,112,112]

    ``generator``: a ``torch.Generator`` (CPU) for the crop / flip draws; default is the global RNG.
    """

    def __init__(self, size=112, mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5), flip_p=0.5):
        self.size, self.big = int(size), int(128 * size / 112)  # train.py:109
        self.flip_p = float(flip_p)
        self.lut_host = torch.from_numpy(normalize_lut(mean, std))
        self._tables = {}  # (device, H, W) -> (xtab, ytab, lut, kx, ky)

    def tables(self, device, h, w):
        key = (str(device), h, w)
        if key not in self._tables:
            xt, yt = resize_tables(h, w, self.big, self.big)
            self._tables[key] = (torch.from_numpy(xt).to(device), torch.from_numpy(yt).to(device),
                                 self.lut_host.to(device), xt.shape[1] - 2, yt.shape[1] - 2)
        return self._tables[key]

    def draw(self, batch, generator=None):
        """(crop int32 [B,2] = (x0, y0), flip uint8 [B]) on the host."""
        span = self.big - self.size + 1
        crop = torch.randint(0, span, (batch, 2), generator=generator, dtype=torch.int32)
        flip = (torch.rand(batch, generator=generator) < self.flip_p).to(torch.uint8)
        return crop, flip

    @staticmethod
    def _to_device(t, dtype, dev):
        """Host draws go through pinned memory with an asynchronous copy: a pageable host-to-device copy would make the
        host wait for everything already queued on the stream, i.e. for the previous training step."""
        t = t.to(dtype).contiguous()
        if t.is_cuda:
            return t
        return t.pin_memory().to(dev, non_blocking=True)

    def __call__(self, u8, crop=None, flip=None, generator=None, validate=True):
        """``validate=False``: the caller guarantees 0 <= crop <= big - size (skips the host-side range check, which is a
        device synchronisation when ``crop`` already lives on the device)."""
        if u8.dtype != torch.uint8 or u8.dim() != 4 or u8.shape[3] != 3:
            raise ValueError("GpuTrainTransform: expected uint8 [B, H, W, 3], got %s %s" % (u8.dtype, tuple(u8.shape)))
        b, h, w, _ = u8.shape
        if b and (h > MAX_ASPECT * w or w > MAX_ASPECT * h):
            raise ValueError("GpuTrainTransform: %dx%d images exceed the supported aspect ratio of %d:1" % (h, w, MAX_ASPECT))
        if crop is None or flip is None:
            crop, flip = self.draw(b, generator)
        span = self.big - self.size
        if b and validate and (int(crop.min()) < 0 or int(crop.max()) > span):
            raise ValueError("GpuTrainTransform: crop offsets must lie in [0, %d]" % span)
        dev = u8.device
        xtab, ytab, lut, kx, ky = self.tables(dev, h, w)
        crop, flip = self._to_device(crop, torch.int32, dev), self._to_device(flip, torch.uint8, dev)
        out = torch.empty(b, 3, self.size, self.size, device=dev, dtype=torch.float32)
        ops.call("fr_augment_u8", u8.contiguous(), xtab, ytab, crop, flip, lut, out, b, h, w, self.big, self.big,
                 self.size, kx, ky, ops.current_stream_ptr())()
        return out
