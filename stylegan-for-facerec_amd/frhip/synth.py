"""Counter-based synthetic tensors (weights, images, labels) that regenerate bit-identically anywhere.

There is no dataset or checkpoint on the GPU box, and the golden fixtures under ``tests/golden`` must not
ship 174 MB of IR-50 weights.  Every tensor is therefore a pure function of ``(seed, name, shape)``:
element ``i`` is ``splitmix64(hash(seed, name) + i)`` mapped to a float.  The generator only uses
integer numpy ops, so it does not depend on any library's RNG stream stability.

Used by ``bench.py`` (synthetic batches), ``tests/`` and ``tests/golden/make_golden.py``.
"""
import zlib

import numpy as np
import torch

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def _stream(seed, name, n):
    base = np.uint64((zlib.crc32(name.encode()) << 32) ^ (int(seed) * 0x9E3779B1 & 0xFFFFFFFF))
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) + _splitmix64(base)
        return _splitmix64(idx)


def uniform(seed, name, shape, lo=-1.0, hi=1.0):
    """float32 tensor, elements i.i.d. U[lo, hi) with 24 random mantissa bits."""
    n = int(np.prod(shape)) if len(shape) else 1
    u = (_stream(seed, name, n) >> np.uint64(40)).astype(np.float64) / float(1 << 24)
    return torch.from_numpy((lo + (hi - lo) * u).astype(np.float32).reshape(shape))


def normal(seed, name, shape, std=1.0):
    """float32 tensor ~ N(0, std^2): sum of 4 uniforms (Irwin-Hall), variance-corrected. Test data only."""
    n = int(np.prod(shape)) if len(shape) else 1
    acc = np.zeros(n, dtype=np.float64)
    for k in range(4):
        acc += (_stream(seed, "%s#%d" % (name, k), n) >> np.uint64(40)).astype(np.float64) / float(1 << 24)
    z = (acc - 2.0) / np.sqrt(4.0 / 12.0)
    return torch.from_numpy((std * z).astype(np.float32).reshape(shape))


def labels(seed, name, n, num_classes):
    """int64 labels uniform in [0, num_classes)."""
    r = _stream(seed, name, n) >> np.uint64(33)
    return torch.from_numpy((r % np.uint64(num_classes)).astype(np.int64))


def fill_state_dict(sd, seed, gain=1.0):
    """Overwrite every tensor of a state dict (in place) with deterministic, *trained-looking* values.

    Shapes decide the distribution so that activations stay O(1) through 50 layers:
      conv / linear weights  U(-b, b), b = gain*sqrt(6/(fan_in+fan_out))   (xavier bound, model_irse.py:174-189)
      BN weight              U(0.8, 1.2);  BN bias U(-0.1, 0.1)
      running_mean           U(-0.1, 0.1); running_var U(0.8, 1.2)
      PReLU slope            U(0.1, 0.4)
      num_batches_tracked    0
    """
    for name, t in sd.items():
        if name.endswith("num_batches_tracked"):
            t.zero_()
            continue
        if t.dim() >= 2:
            rf = int(np.prod(t.shape[2:])) if t.dim() > 2 else 1
            fan_in, fan_out = t.shape[1] * rf, t.shape[0] * rf
            b = gain * float(np.sqrt(6.0 / (fan_in + fan_out)))
            v = uniform(seed, name, tuple(t.shape), -b, b)
        elif name.endswith("running_var"):
            v = uniform(seed, name, tuple(t.shape), 0.8, 1.2)
        elif name.endswith("running_mean"):
            v = uniform(seed, name, tuple(t.shape), -0.1, 0.1)
        elif name.endswith(".bias"):
            v = uniform(seed, name, tuple(t.shape), -0.1, 0.1)
        else:
            # 1-D ".weight": BN gamma or PReLU slope.  PReLU sits at index 2 of its Sequential in both
            # the stem (input_layer.2) and the residual branch (res_layer.2).
            if name.endswith("input_layer.2.weight") or name.endswith("res_layer.2.weight"):
                v = uniform(seed, name, tuple(t.shape), 0.1, 0.4)
            else:
                v = uniform(seed, name, tuple(t.shape), 0.8, 1.2)
        t.copy_(v.to(t.dtype))
    return sd
